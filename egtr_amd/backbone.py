"""ResNet-50 feature backbone with frozen batch-norm, plain torch.nn (runs on PyTorch-ROCm / MIOpen).

The reference wraps ``timm.create_model("resnet50", features_only=True, out_indices=(2, 3, 4))`` and replaces every
BatchNorm2d by a frozen variant (model/deformable_detr.py:666-787).  timm is not available here, so the same
architecture is defined directly with timm's parameter names (``conv1``, ``bn1``, ``layer{1..4}.{b}.conv{1,2,3}``,
``bn{1,2,3}``, ``downsample.{0,1}``) so that a reference checkpoint's
``model.backbone.conv_encoder.model.*`` keys load unchanged.  Out of scope for hand-written kernels per the
north-star ("host code stays Python on PyTorch-ROCm for the ResNet-50 backbone").
"""
import torch
from torch import nn


class DeformableDetrFrozenBatchNorm2d(nn.Module):
    """Fixed statistics and affine parameters, eps = 1e-5 added before rsqrt (dd:666-714)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)  # dd:690-692
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def forward(self, x):
        scale = self.weight * (self.running_var + 1e-5).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = DeformableDetrFrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = DeformableDetrFrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
        self.bn3 = DeformableDetrFrozenBatchNorm2d(out)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride=stride, bias=False),
                                            DeformableDetrFrozenBatchNorm2d(out))
        else:
            self.downsample = None

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return torch.relu(y + idt)


class ResNet50Features(nn.Module):
    """Returns the C3, C4, C5 maps (strides 8, 16, 32; 512, 1024, 2048 channels)."""

    channels = [512, 1024, 2048]
    reductions = [8, 16, 32]

    def __init__(self, out_indices=(2, 3, 4)):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = DeformableDetrFrozenBatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layers = []
            for b in range(blocks):
                layers.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, downsample=(b == 0)))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layers))

    def forward(self, x):
        x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
        feats = []
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
            if li in self.out_indices:
                feats.append(x)
        return feats
