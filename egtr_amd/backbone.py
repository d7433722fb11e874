"""ResNet-50 feature backbone with frozen batch-norm, plain torch.nn (runs on PyTorch-ROCm / MIOpen).

The reference wraps ``timm.create_model("resnet50", features_only=True, out_indices=(2, 3, 4))`` and replaces every
BatchNorm2d by a frozen variant (model/deformable_detr.py:666-787).  timm is not available here, so the same
architecture is defined directly with timm's parameter names (``conv1``, ``bn1``, ``layer{1..4}.{b}.conv{1,2,3}``,
``bn{1,2,3}``, ``downsample.{0,1}``) so that a reference checkpoint's
``model.backbone.conv_encoder.model.*`` keys load unchanged.  Out of scope for hand-written kernels per the
north-star ("host code stays Python on PyTorch-ROCm for the ResNet-50 backbone").
"""
import os

import torch
import torch.nn.functional as F
from torch import nn


TRAIN_FUSED_EPILOGUE = True   # module attribute (tests patch it); no environment switch since round 6
# bf16 inference: the backbone behind the stem runs channels-last -- MIOpen's NHWC 3x3 convolutions on the tensors as they lie
# (on NCHW tensors the same kernels run between two layout transposes: 418 vs 176 us for a layer-1 convolution at bs 16,
# tools/nhwc_probe.py) and the 1x1 convolutions as plain [N*H*W, Cin] x [Cin, Cout] GEMMs with bias (+ ReLU) in the epilogue.
# EGTR_BACKBONE_NHWC=0 (one switch for both dtypes since round 6): NCHW throughout.
NHWC_BF16 = os.environ.get("EGTR_BACKBONE_NHWC", "1") != "0"
# fp32 inference: the same layout behind the (NCHW, fused pool) stem -- MIOpen's NHWC 3x3 kernels measure 10-30 % faster than
# its NCHW choices at bs 1 (tools/nhwc_probe.py --fp32: 786 -> 660 us over the 16 convolutions) and the conv1 epilogue
# launches disappear into the GEMMs.
NHWC_F32 = os.environ.get("EGTR_BACKBONE_NHWC", "1") != "0"
# fp32 channels-last bottleneck: shift + ReLU of conv2, conv3, shift + shortcut + ReLU as one launch of the bottleneck-tail
# kernel (Bottleneck.forward_folded_nhwc, csrc/conv_tail_x6.hip).  EGTR_GEMM_SPLIT_BF16=0 turns this off together with the other
# split-bf16 routes.
CONV3_FUSED = True        # module attributes (tests patch them for the switch-off twins)
CONV3_FUSED_BF16 = True   # the bf16 twin (csrc/conv_tail_bf16.hip)
CONV1_X6 = True           # fp32 first 1x1 convolution of a block through the tail kernel ...
CONV1_X6_SHAPES = ((64, 64),)   # ... for these (input channels, planes): 7 us against the vendor GEMM's 16 at 64 -> 64 (layer 1,
                          # block 0); at 256 -> 64 and wider the two are level inside the forward (tools/conv2_ab.sh, SWITCH=CONV1_X6)
SHORTCUT_X6 = True        # fp32 stride-2 shortcut projections as the one-tap form of csrc/conv3x3_x6.hip
FROZEN_PREFIX_NHWC = True  # training: the frozen stem + layer 1 through the channels-last inference kernels
STEM_FUSED_BF16 = True    # the bf16 twin (csrc/stem_bf16.hip)
STEM_FUSED = True         # fp32 stem: 7x7 convolution + shift + ReLU + max-pool as csrc/stem_x6.hip
CONV2_X6 = True           # fp32 3x3 convolutions as csrc/conv3x3_x6.hip ...
CONV2_X6_MAX_WIDTH = 512  # ... up to this width (all 16 of ResNet-50)


def _fold(conv, bn):
    """Fold a frozen BN into the preceding conv: y = conv(x, w * scale) + shift (exact algebra; fp32 rounding differs
    from scale-after-conv by ~1e-7 relative).  The folding arithmetic is fp32 whatever the model dtype; the folded
    weight is stored in the model dtype, the shift always in fp32 (it is applied by the fused epilogue kernel)."""
    scale = bn.weight.float() * (bn.running_var.float() + 1e-5).rsqrt()
    shift = bn.bias.float() - bn.running_mean.float() * scale
    return ((conv.weight.float() * scale.reshape(-1, 1, 1, 1)).to(conv.weight.dtype).contiguous(),
            shift.contiguous())


def conv1x1_as_gemm(x, weight):
    """A stride-1, unpadded 1x1 convolution on NCHW data IS the GEMM  Y[b] = W[Cout, Cin] . X[b][Cin, H*W]  on the
    tensor as it lies in memory (no layout change).  Issued through torch.matmul it is served by the rocBLAS / hipBLASLt
    solution TunableOp picked for that shape (egtr_amd.runtime.enable_gemm_tuning) instead of MIOpen's own untuned
    rocBLAS call: ResNet-50 feature extraction at 600x1000 2.03 -> 1.85 ms (tools/backbone_probe.py, variant d)."""
    b, c, h, w_ = x.shape
    return torch.matmul(weight.reshape(weight.shape[0], c), x.reshape(b, c, h * w_)).view(b, -1, h, w_)


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


class ScaleWeightsFunction(torch.autograd.Function):
    """w_i * scale_i for ALL trainable convolutions of the backbone in one multi-tensor launch (the frozen BN's scale riding on
    the convolution weight, Bottleneck.forward_train_fused), and their gradients scaled back in one more: 42 + 42 launches per
    train step otherwise (aten::mul was the most frequent ATen kernel of the step, tools/train_glue_sources.py)."""

    @staticmethod
    def forward(ctx, scales, *weights):
        from . import ops
        ctx.scales = scales
        return tuple(ops.scale_rows_multi([w.detach() for w in weights], scales))

    @staticmethod
    def backward(ctx, *grads):
        from . import ops
        present = [i for i, g in enumerate(grads) if g is not None]
        out = [None] * len(grads)
        if present:
            scaled = ops.scale_rows_multi([grads[i] for i in present], [ctx.scales[i] for i in present])
            for i, v in zip(present, scaled):
                out[i] = v
        return (None, *out)


SCALE_WEIGHTS_FUSED = True   # module attribute (tests patch it); no environment switch since round 6


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

    def train_fused_applies(self, x):
        return (TRAIN_FUSED_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and self.conv1.weight.dtype == torch.float32)

    def forward(self, x, scaled=None):
        # (training path: the 1x1 convolutions stay on MIOpen -- as torch.matmul their backward through the batch
        # broadcast was measured slower, 86.8 vs 68.2 ms per bs=4 train step)
        if self.train_fused_applies(x):
            return self.forward_train_fused(x, scaled)
        idt = x if self.downsample is None else self.downsample(x)
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return torch.relu(y + idt)

    def train_affine(self):
        """[(scale [C,1,1,1], shift [C])] of bn1, bn2, bn3 (, downsample BN, (None, shift3 + shift_d)): cached per buffer version."""
        from . import ops
        bns = [self.bn1, self.bn2, self.bn3] + ([self.downsample[1]] if self.downsample is not None else [])

        def build():
            out = []
            for bn in bns:
                scale = bn.weight.float() * (bn.running_var.float() + 1e-5).rsqrt()
                out.append((scale.reshape(-1, 1, 1, 1).contiguous(),
                            (bn.bias.float() - bn.running_mean.float() * scale).contiguous()))
            if self.downsample is not None:
                out.append((None, (out[2][1] + out[3][1]).contiguous()))   # the shortcut's shift rides with conv3's
            return out

        with torch.no_grad():
            return ops.cached_weights(self, "bn_affine_train",
                                      [t for bn in bns for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)], build)

    def train_convs(self):
        return [self.conv1, self.conv2, self.conv3] + ([self.downsample[0]] if self.downsample is not None else [])

    def forward_train_fused(self, x, scaled=None):
        """Training with trainable convolutions (layers 2-4): the frozen BN's scale multiplies the convolution WEIGHT under
        autograd (a [Cout, Cin, k, k] product instead of a pass over the activation; d loss / d weight picks the scale
        up through that product), and shift + residual + ReLU are one HIP pass forward (egtr_bias_act_nchw_f32) and one
        mask pass backward -- instead of rsqrt / mul / sub on the statistics (5 launches per BN, every step), a
        multiplication and an addition over the activation, the residual add, the ReLU and their three backward passes.
        Same algebra as the folded inference path (fp32 rounding differs from scale-after-conv by ~1e-7 relative)."""
        from . import ops
        c = self.train_affine()
        if scaled is None:   # (one block on its own: the per-weight products; ResNet50Features hands in all blocks' at once)
            scaled = [cv.weight * c[i][0] for i, cv in enumerate(self.train_convs())]
        idt, shift3 = x, c[2][1]
        if self.downsample is not None:
            ds = self.downsample[0]
            idt = F.conv2d(x, scaled[3], None, stride=ds.stride)
            shift3 = c[4][1]
        y = ops.bias_act(F.conv2d(x, scaled[0]), c[0][1])
        y = ops.bias_act(F.conv2d(y, scaled[1], None, stride=self.conv2.stride, padding=1), c[1][1])
        return ops.bias_act(F.conv2d(y, scaled[2]), shift3, idt)

    def folded_params(self):
        p = [_fold(self.conv1, self.bn1), _fold(self.conv2, self.bn2), _fold(self.conv3, self.bn3)]
        if self.downsample is not None:
            p.append(_fold(self.downsample[0], self.downsample[1]))
            # the shortcut's shift is applied together with conv3's (one epilogue pass instead of two)
            p.append((None, (p[2][1] + p[3][1]).contiguous()))
        return p

    def folded_params_nhwc(self):
        """Folded weights for the channels-last bf16 path: 1x1 convolutions as [Cout, Cin] matrices (+ their shift as a bf16 bias
        for the GEMM epilogue where a ReLU follows directly), the 3x3 as a channels-last weight; shifts in fp32 for the epilogue
        kernel."""
        p = self.folded_params()
        w1, b1 = p[0]
        w2, b2 = p[1]
        w3, b3 = p[2]
        out = {"w1": w1.reshape(w1.shape[0], -1).contiguous(), "b1": b1.to(w1.dtype).contiguous(),
               "w2": w2.contiguous(memory_format=torch.channels_last), "b2": b2,
               "w3": w3.reshape(w3.shape[0], -1).contiguous(), "b3": b3, "wd": None}
        if self.downsample is not None:
            wd = p[3][0]
            out["wd"] = (wd.reshape(wd.shape[0], -1).contiguous() if tuple(self.downsample[0].stride) == (1, 1)
                         else wd.contiguous(memory_format=torch.channels_last))
            out["b3"] = p[4][1]     # the shortcut's shift rides along with conv3's
        return out

    def forward_folded_nhwc(self, x, q):
        """Inference, x a channels-last [B, C, H, W] tensor: conv1 = GEMM + bias + ReLU in its epilogue (no pass of its own),
        conv2 = MIOpen NHWC; then fp32: the tail kernel (conv2's shift + ReLU, conv3, shift + shortcut + ReLU in one launch);
        bf16: one epilogue pass, conv3 = GEMM, shift + shortcut + ReLU in one pass."""
        from . import ops
        B, C, H, W_ = x.shape
        x2 = x.permute(0, 2, 3, 1).reshape(-1, C)                      # a view: channels-last IS [B*H*W, C]
        N1 = q["w1"].shape[0]
        if (CONV1_X6 and ops.GEMM_SPLIT_BF16 and (C, N1) in CONV1_X6_SHAPES and ops.conv1x1_tail_supported(x2, N1)):
            # fp32, block inputs of up to 512 channels: the panel-resident split-bf16 kernel of the tail, here without input
            # shift and shortcut (csrc/conv_tail_x6.hip), instead of the vendor GEMM with its bias + ReLU epilogue
            if "w1xs" not in q:
                q["w1xs"] = ops.xs_split(q["w1"], weights=True)
                q["b1f"] = q["b1"].float().contiguous()
            y = ops.conv1x1_tail(x2, None, q["w1xs"], q["b1f"], None, N1, relu_in=False, relu_out=True)
        else:
            y = torch._addmm_activation(q["b1"], x2, q["w1"].t(), use_gelu=False)
        y = y.view(B, H, W_, -1).permute(0, 3, 1, 2)                   # channels-last view of the GEMM's output
        st2 = tuple(self.conv2.stride)
        if (CONV2_X6 and ops.GEMM_SPLIT_BF16 and st2 in ((1, 1), (2, 2)) and q["w2"].shape[0] <= CONV2_X6_MAX_WIDTH
                and ops.conv3x3_supported(y, q["w2"].shape[0], st2[0])):
            # fp32: the 3x3 convolution as a split-bf16 implicit GEMM of our own (csrc/conv3x3_x6.hip) -- 22 us where MIOpen's
            # fp32-MFMA kernels take 41 at 600 x 1000
            if "w2xs" not in q:
                q["w2xs"] = ops.conv3x3_weights(q["w2"], st2[0])
            y = ops.conv3x3(y, q["w2xs"], q["w2"].shape[0], st2[0])
        else:
            y = F.conv2d(y, q["w2"], None, stride=self.conv2.stride, padding=1)
        if not y.is_contiguous(memory_format=torch.channels_last):
            # PYTORCH_MIOPEN_SUGGEST_NHWC was not in effect when this process ran its FIRST convolution (it is read once; the
            # package sets it at import, see egtr_amd/__init__.py): the convolution ran NCHW between two layout transposes
            # and the reshape below copies.  Correct, slower -- and never silent.
            ops.note_fallback("backbone_nhwc", "a channels-last convolution returned an NCHW tensor: "
                              "PYTORCH_MIOPEN_SUGGEST_NHWC=1 was not set before the process's first convolution")
        Ho, Wo = y.shape[-2:]
        y2 = y.permute(0, 2, 3, 1).reshape(-1, y.shape[1])
        N3 = q["w3"].shape[0]
        if self.downsample is None:
            idt = x2
        elif q["wd"].dim() == 2:
            idt = torch.mm(x2, q["wd"].t())
        elif (SHORTCUT_X6 and ops.GEMM_SPLIT_BF16 and tuple(self.downsample[0].stride) == (2, 2)
              and ops.conv1x1_strided_supported(x, N3, 2)):
            # fp32: the stride-2 shortcut projection through the one-tap form of the own convolution kernel (csrc/conv3x3_x6.hip)
            if "wdxs" not in q:
                q["wdxs"] = ops.xs_split(q["wd"].reshape(N3, C).contiguous(), weights=True)
            idt = ops.conv1x1_strided(x, q["wdxs"], N3, 2)
        else:
            idt = F.conv2d(x, q["wd"], None, stride=self.downsample[0].stride).permute(0, 2, 3, 1).reshape(-1, N3)
        if (CONV3_FUSED and ops.GEMM_SPLIT_BF16 and ops.conv1x1_tail_supported(y2, N3) and idt.dtype == torch.float32
                and idt.stride(1) == 1 and idt.stride(0) % 4 == 0 and idt.data_ptr() % 16 == 0):
            # fp32: the block's tail as ONE launch (csrc/conv_tail_x6.hip, fp32-accurate split-bf16 products): conv2's shift +
            # ReLU applied to the rows on their way into LDS, conv3's shift + shortcut + ReLU in the epilogue -- instead of
            # pass, vendor GEMM, pass (two round trips of the activation through memory and two launches less per block)
            if "w3xs" not in q:
                q["w3xs"] = ops.xs_split(q["w3"], weights=True)
            z = ops.conv1x1_tail(y2, q["b2"], q["w3xs"], q["b3"], idt, N3)
            return z.view(B, Ho, Wo, -1).permute(0, 3, 1, 2)
        if (CONV3_FUSED_BF16 and ops.conv1x1_tail_bf16_supported(y2, N3) and idt.dtype == torch.bfloat16 and idt.stride(1) == 1
                and idt.stride(0) % 8 == 0 and idt.data_ptr() % 16 == 0):
            # bf16: the same as one launch of csrc/conv_tail_bf16.hip, with the rounding points of the composition below
            if "w3pk" not in q:
                q["w3pk"] = ops.conv_tail_pack_bf16(q["w3"])
            z = ops.conv1x1_tail_bf16(y2, q["b2"], q["w3pk"], q["b3"], idt, N3)
            return z.view(B, Ho, Wo, -1).permute(0, 3, 1, 2)
        ops.bias_act_rows_(y2, q["b2"])
        z = torch.mm(y2, q["w3"].t())
        ops.bias_act_rows_(z, q["b3"], idt)
        return z.view(B, Ho, Wo, -1).permute(0, 3, 1, 2)

    def forward_folded(self, x, p):
        """Inference path: frozen BN folded into the conv weights; the per-channel shift, the residual add and the ReLU
        are ONE fused HIP pass after each convolution (csrc/elementwise.hip).  (torch.miopen_convolution_relu was
        tried first: for fp32 it still runs separate broadcast-add and clamp kernels -- rocprofv3, round 1.)"""
        from . import ops
        s = self.conv2.stride
        idt = x
        shift3 = p[2][1]
        if self.downsample is not None:
            # raw shortcut convolution: its frozen-BN shift rides along with conv3's in the final epilogue pass,
            # relu(conv3 + (shift3 + shift_d) + shortcut) -- one pass over the block output instead of two
            if tuple(self.downsample[0].stride) == (1, 1):
                idt = conv1x1_as_gemm(x, p[3][0])
            else:
                idt = F.conv2d(x, p[3][0], None, stride=self.downsample[0].stride)
            shift3 = p[4][1]
        y = ops.bias_act_(conv1x1_as_gemm(x, p[0][0]), p[0][1])
        y = ops.bias_act_(F.conv2d(y, p[1][0], None, stride=s, padding=1), p[1][1])
        return ops.bias_act_(conv1x1_as_gemm(y, p[2][0]), shift3, idt)


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

        self._folded = None
        self._folded_key = None
        self._frozen_folded = None
        self._frozen_key = None

    def _frozen_prefix(self):
        """Stem + layer1 when none of their parameters takes gradients (the reference freezes exactly these:
        dd:763-770) -- nothing upstream of them is trainable either, so in training they can run without autograd."""
        mods = [self.conv1, self.layer1]
        params = [p for m in mods for p in m.parameters()]
        return params if params and not any(p.requires_grad for p in params) else None

    def _forward_frozen_prefix(self, x, params):
        """Stem + layer1 through the folded-BN fused path under no_grad (training steps: 3-4 elementwise passes over the
        largest activations of the network per convolution become one, and nothing is saved for a backward that never
        runs).  Folded weights cached on the versions of the frozen parameters only."""
        from . import ops
        key = tuple(p._version for p in params) + (str(x.device), x.dtype, self.conv1.weight.data_ptr())
        with torch.no_grad():
            if self._frozen_folded is None or key != self._frozen_key:
                self._frozen_folded = {"stem": _fold(self.conv1, self.bn1),
                                       1: [blk.folded_params() for blk in self.layer1]}
                self._frozen_key = key
            w, b = self._frozen_folded["stem"]
            mp = self.maxpool
            if (FROZEN_PREFIX_NHWC and NHWC_F32 and STEM_FUSED and ops.GEMM_SPLIT_BF16 and x.dtype == torch.float32
                    and ops.stem_fused_supported(x, w) and hasattr(torch, "_addmm_activation")
                    and tuple(self.conv1.stride) == (2, 2) and tuple(self.conv1.padding) == (3, 3) and mp.kernel_size == 3
                    and mp.stride == 2 and mp.padding == 1 and mp.dilation == 1 and not mp.ceil_mode):
                # the inference kernels of the channels-last route (fused stem, own 3x3 convolutions, bottleneck tails) for the
                # part of the network that never takes gradients; the trainable layers behind it run NCHW: one layout change
                if "nhwc" not in self._frozen_folded:
                    self._frozen_folded["nhwc"] = [blk.folded_params_nhwc() for blk in self.layer1]
                    self._frozen_folded["stem_xs"] = ops.stem_weights(w)
                x = ops.stem_fused(x, self._frozen_folded["stem_xs"], b)
                for blk, q in zip(self.layer1, self._frozen_folded["nhwc"]):
                    x = blk.forward_folded_nhwc(x, q)
                return x.contiguous()
            x = self._stem_folded(x, w, b)
            for blk, p in zip(self.layer1, self._frozen_folded[1]):
                x = blk.forward_folded(x, p)
        return x

    def _stem_folded(self, x, w, b):
        """maxpool(relu(conv + shift)) == relu(maxpool(conv) + shift) bit for bit (a per-channel constant and monotone
        rounding commute with max; the pool pads with -inf): the shift / ReLU pass runs on the pooled tensor, a quarter
        of the convolution output."""
        from . import ops
        y = F.conv2d(x, w, None, stride=2, padding=3)
        mp = self.maxpool
        if (y.dtype == torch.float32 and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1
                and mp.dilation == 1 and not mp.ceil_mode and y.shape[0] * y.shape[1] <= 65535):
            return ops.bias_relu_maxpool(y, b)  # pool + shift + ReLU in one pass over the convolution output
        return ops.bias_act_(mp(y), b)

    def _fold_key(self):
        """What the folded weights depend on: parameters AND the frozen-BN buffers (load_state_dict copies into both in place).
        The full key (version counter of all 265 tensors) costs ~0.3 ms of host time, and the eager forward is host-bound:
        it is recomputed when one of a few SENTINELS moved -- the stem and the last trainable convolution weight, one
        mid-network weight, one BatchNorm buffer (an optimizer step, load_state_dict, .to() move all of them or their storage)
        -- and every 64th call as a backstop for an in-place edit of some other single tensor; ops.invalidate_derived resets."""
        d = self.__dict__
        w0, w1, w2, bv = self.conv1.weight, self.layer4[-1].conv3.weight, self.layer2[0].conv1.weight, self.bn1.running_var
        quick = (w0._version, w1._version, w2._version, bv._version, w0.data_ptr(), w1.data_ptr(), bv.data_ptr(), w0.dtype,
                 str(w0.device))
        n = d.get("_fold_calls", 0) + 1
        d["_fold_calls"] = n
        if d.get("_fold_quick") == quick and d.get("_fold_full") is not None and n % 64:
            return d["_fold_full"]
        ts = list(self.parameters()) + list(self.buffers())
        full = tuple([t._version for t in ts]) + quick
        d["_fold_quick"], d["_fold_full"] = quick, full
        return full

    def forward(self, x):
        # Inference (no grad, eval, GPU): folded-BN + fused conv/bias/ReLU path; the folded weights are cached and
        # rebuilt if any parameter was modified in place (optimizer step, load_state_dict).
        if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and self.conv1.weight.dtype == x.dtype
                and not torch.is_grad_enabled() and not self.training):
            key = self._fold_key()
            if self._folded is None or key != self._folded_key:
                with torch.no_grad():
                    self._folded = {"stem": _fold(self.conv1, self.bn1)}
                    for li in range(1, 5):
                        self._folded[li] = [blk.folded_params() for blk in getattr(self, f"layer{li}")]
                self._folded_key = key
            from . import ops
            w, b = self._folded["stem"]
            feats = []
            if (((NHWC_BF16 and x.dtype == torch.bfloat16) or (NHWC_F32 and x.dtype == torch.float32))
                    and hasattr(torch, "_addmm_activation")):
                if "nhwc" not in self._folded:
                    with torch.no_grad():
                        self._folded["nhwc"] = {li: [blk.folded_params_nhwc() for blk in getattr(self, f"layer{li}")]
                                                for li in range(1, 5)}
                        self._folded["nhwc"]["stem"] = w.contiguous(memory_format=torch.channels_last)
                mp0 = self.maxpool
                stem_std = (tuple(self.conv1.stride) == (2, 2) and tuple(self.conv1.padding) == (3, 3) and mp0.kernel_size == 3
                            and mp0.stride == 2 and mp0.padding == 1 and mp0.dilation == 1 and not mp0.ceil_mode)
                if x.dtype == torch.bfloat16 and STEM_FUSED_BF16 and stem_std and ops.stem_fused_bf16_supported(x, w):
                    # bf16: convolution + shift + ReLU + pool in one launch, channels-last out (csrc/stem_bf16.hip)
                    if "stem_pk" not in self._folded["nhwc"]:
                        self._folded["nhwc"]["stem_pk"] = ops.stem_weights_bf16(w)
                    x = ops.stem_fused_bf16(x, self._folded["nhwc"]["stem_pk"], b)
                elif x.dtype == torch.bfloat16:
                    # channels-last from the pixels on: the stem convolution and the pool on channels-last tensors as well (the
                    # stem ran MIOpen's NHWC kernel between two layout transposes anyway), shift + ReLU on the pooled tensor
                    x = F.conv2d(x.contiguous(memory_format=torch.channels_last), self._folded["nhwc"]["stem"], None, stride=2,
                                 padding=3)
                    x = self.maxpool(x).contiguous(memory_format=torch.channels_last)    # (already channels-last: no copy)
                    ops.bias_act_rows_(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]), b)
                else:
                    # fp32: the stem keeps its one-pass pool + shift + ReLU kernel (NCHW); the pooled map changes layout once
                    mp = self.maxpool
                    if (STEM_FUSED and ops.GEMM_SPLIT_BF16 and ops.stem_fused_supported(x, w) and tuple(self.conv1.stride) == (2, 2)
                            and tuple(self.conv1.padding) == (3, 3) and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1
                            and mp.dilation == 1 and not mp.ceil_mode):
                        # convolution + shift + ReLU + pool in one launch, channels-last out (csrc/stem_x6.hip)
                        if "stem_xs" not in self._folded["nhwc"]:
                            self._folded["nhwc"]["stem_xs"] = ops.stem_weights(w)
                        x = ops.stem_fused(x, self._folded["nhwc"]["stem_xs"], b)
                    else:
                        x = self._stem_folded(x, w, b).contiguous(memory_format=torch.channels_last)
                for li in range(1, 5):
                    for blk, q in zip(getattr(self, f"layer{li}"), self._folded["nhwc"][li]):
                        x = blk.forward_folded_nhwc(x, q)
                    if li in self.out_indices:
                        feats.append(x)       # channels-last: DeformableDetrModel projects them as token matrices
                return feats
            x = self._stem_folded(x, w, b)
            for li in range(1, 5):
                for blk, p in zip(getattr(self, f"layer{li}"), self._folded[li]):
                    x = blk.forward_folded(x, p)
                if li in self.out_indices:
                    feats.append(x)
            return feats
        frozen = self._frozen_prefix() if (x.is_cuda and x.dtype == torch.float32 and not x.requires_grad
                                           and self.conv1.weight.dtype == x.dtype) else None
        feats = []
        if frozen is not None:
            x = self._forward_frozen_prefix(x, frozen)
            if 1 in self.out_indices:
                feats.append(x)
            first = 2
        else:
            x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
            first = 1
        blocks = [blk for li in range(first, 5) for blk in getattr(self, f"layer{li}")]
        scaled = None
        if SCALE_WEIGHTS_FUSED and blocks and all(b.train_fused_applies(x) for b in blocks):
            # every trainable convolution weight times its frozen-BN scale: one multi-tensor launch (and one for the gradients)
            ws = [cv.weight for b in blocks for cv in b.train_convs()]
            sc = [b.train_affine()[i][0] for b in blocks for i in range(len(b.train_convs()))]
            flat = ScaleWeightsFunction.apply(sc, *ws)
            scaled, o = [], 0
            for b in blocks:
                n = len(b.train_convs())
                scaled.append(list(flat[o:o + n]))
                o += n
        bi = 0
        for li in range(first, 5):
            for blk in getattr(self, f"layer{li}"):
                x = blk(x, scaled[bi]) if scaled is not None else blk(x)
                bi += 1
            if li in self.out_indices:
                feats.append(x)
        return feats
