"""Import shims that let the read-only reference at /root/reference be imported in THIS container.

Test infrastructure only (golden-vector generation). Contains no reference code: it only patches the
import environment (transformers 5.x removed a few names the reference imports; torchvision / timm are
absent) and then does ``sys.path.insert(0, "/root/reference")``.  Inert where /root/reference is absent
(the GPU box) -- ``load_reference()`` raises FileNotFoundError there.

Shim list follows SURVEY.md section 8(c).
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def load_reference():
    if not os.path.isdir(REF_ROOT):
        raise FileNotFoundError(REF_ROOT)
    import torch
    import transformers
    import transformers.modeling_utils as mu
    from transformers import PretrainedConfig

    # 1. transformers.DetrFeatureExtractor was removed upstream (reference: model/deformable_detr.py:45)
    if not hasattr(transformers, "DetrFeatureExtractor"):
        class _DetrFeatureExtractor:  # noqa: D401 - dummy base, preprocessing is out of scope
            def __init__(self, *a, **k):
                pass
        transformers.DetrFeatureExtractor = _DetrFeatureExtractor
    # 2. transformers.modeling_utils.PretrainedConfig (model/deformable_detr.py:57)
    if not hasattr(mu, "PretrainedConfig"):
        mu.PretrainedConfig = PretrainedConfig
    # 3. transformers.models.detr.feature_extraction_detr.center_to_corners_format
    #    (model/egtr.py:35, model/deformable_detr.py:461-464)
    modname = "transformers.models.detr.feature_extraction_detr"
    if modname not in sys.modules:
        m = types.ModuleType(modname)

        def center_to_corners_format(x):
            x_c, y_c, w, h = x.unbind(-1)
            b = [(x_c - 0.5 * w), (y_c - 0.5 * h), (x_c + 0.5 * w), (y_c + 0.5 * h)]
            return torch.stack(b, dim=-1)

        m.center_to_corners_format = center_to_corners_format
        sys.modules[modname] = m
    # transformers.file_utils names used by model/deformable_detr.py:46-55
    import transformers.file_utils as fu
    import transformers.utils as tu
    for name in ("ModelOutput", "add_start_docstrings", "is_scipy_available", "is_timm_available",
                 "is_torch_cuda_available", "is_vision_available", "requires_backends"):
        if not hasattr(fu, name):
            if hasattr(tu, name):
                setattr(fu, name, getattr(tu, name))
            elif name == "add_start_docstrings":
                setattr(fu, name, lambda *a, **k: (lambda f: f))
            elif name == "is_torch_cuda_available":
                setattr(fu, name, lambda: torch.cuda.is_available())
            else:
                raise ImportError(name)
    # 4. model.transform needs torchvision (absent): data augmentation only, never on the hot path
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "model.transform" not in sys.modules:
        import model  # the reference's (empty) package  # noqa: F401
        sys.modules["model.transform"] = types.ModuleType("model.transform")
    import model.deformable_detr as dd
    import model.egtr as eg
    return dd, eg


def make_stub_backbone_class():
    """Stand-in for DeformableDetrTimmConvEncoder (timm absent): three 1x1 convs on strided
    views giving C3/C4/C5-shaped maps (512/1024/2048 channels at strides 8/16/32, ceil sizes)."""
    import torch
    from torch import nn

    class StubBackbone(nn.Module):
        def __init__(self, config=None, channels=(512, 1024, 2048)):
            super().__init__()
            self.intermediate_channel_sizes = list(channels)
            self.strides = [8, 16, 32]
            self.model = nn.ModuleList([nn.Conv2d(3, c, kernel_size=1) for c in channels])

        def forward(self, pixel_values, pixel_mask):
            out = []
            for conv, s in zip(self.model, self.strides):
                f = conv(pixel_values[:, :, ::s, ::s])
                mask = nn.functional.interpolate(pixel_mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
                out.append((f, mask))
            return out

    return StubBackbone
