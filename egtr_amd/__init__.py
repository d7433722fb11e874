"""egtr_amd -- MI355X (gfx950) native hot path of EGTR scene-graph generation.

Layout:
  csrc/               hand-written HIP kernels + the C ABI (include/egtr_hip.h) -> libegtr_hip.so
  _lib.py             ctypes binding of that C ABI (fails loudly if the library is missing)
  load_custom.py      drop-in for the reference's model/load_custom.py: returns an object exposing
                      ms_deform_attn_forward / ms_deform_attn_backward with the reference's pybind signatures
  ops.py              autograd bridges (MSDA, decoder self-attention core, relation head)
  deformable_detr.py  host-side mirror of the reference's model/deformable_detr.py module API
  egtr.py             host-side mirror of model/egtr.py (DetrForSceneGraphGeneration, SceneGraphGenerationLoss)
  util.py             loss helpers (mirror of model/util.py)
  runtime.py          one-process-per-GPU data-parallel helpers (RCCL via torch.distributed), HIP-graph capture
"""
import os as _os

# The inference backbone runs channels-last in bf16 AND fp32 (egtr_amd.backbone, NHWC_BF16 / NHWC_F32): PyTorch-ROCm hands
# channels-last tensors to MIOpen's NHWC convolutions only with this switch, read ONCE at the process's first convolution -- so
# it is set at import, before any.
# SIDE EFFECT, process-wide: every other model in this process that feeds channels-last tensors to a convolution gets MIOpen's
# NHWC kernels too (contiguous NCHW tensors are not affected); export PYTORCH_MIOPEN_SUGGEST_NHWC=0 before the import to keep
# PyTorch's default -- the backbone then announces the slower route once (ops.note_fallback "backbone_nhwc"), as it does when a
# convolution already ran before this import and the switch therefore came too late.
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

__version__ = "0.1.0"
