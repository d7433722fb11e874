"""Drop-in replacement for the reference's kernel loader (model/load_custom.py:23-57).

The reference JIT-compiles its CUDA sources with torch.utils.cpp_extension.load() and returns a pybind module
exposing ``ms_deform_attn_forward`` / ``ms_deform_attn_backward`` (model/custom_kernel/vision.cpp:13-15).
Here the kernels are hand-written HIP, prebuilt in-tree into libegtr_hip.so (C ABI, include/egtr_hip.h), and this
module returns an object with the SAME two functions and the same argument lists, so
``MultiScaleDeformableAttentionFunction`` (model/deformable_detr.py:402-455) can call it unchanged.

Error convention follows the reference's AT_ASSERTM host checks (ms_deform_attn_cuda.cu:31-41, 96-108): a
RuntimeError for non-contiguous / non-device operands.  Unlike the reference there is no silent PyTorch
fallback (deformable_detr.py:1096-1101 swallows every exception): a failure here propagates.
"""
import torch

from . import _lib


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.data_ptr() % 16 != 0:
        raise RuntimeError(f"{name} must be 16-byte aligned")
    return t


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The raw handle of torch's current stream on the current device.  Every launch through the C ABI asks for it (~60 per
    eager forward): the private raw accessor answers in ~0.3 us, ``torch.cuda.current_stream().cuda_stream`` builds a Stream
    object first (~6 us: 0.4 ms of host time per forward, and the eager forward is host-bound)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def pack_keep_bits(mask, B, S):
    """[B, S] padding mask (any dtype; non-zero = real token) -> [B, ceil(S/32)] int32, bit t%32 of word t/32 = token t."""
    words = (S + 31) // 32
    m = torch.zeros(B, words * 32, dtype=torch.int64, device=mask.device)
    m[:, :S] = (mask.reshape(B, S) != 0).to(torch.int64)
    w = (m.view(B, words, 32) << torch.arange(32, device=mask.device, dtype=torch.int64)).sum(-1)
    return ((w + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32).contiguous()


def _check_keep_bits(bits, B, S):
    if (not torch.is_tensor(bits) or not bits.is_cuda or bits.dtype != torch.int32
            or tuple(bits.shape) != (B, (S + 31) // 32) or not bits.is_contiguous()):
        raise RuntimeError(f"keep_bits must be a contiguous int32 CUDA/HIP tensor of shape {(B, (S + 31) // 32)}")
    return bits


class _MultiScaleDeformableAttention:
    """Module-like object bound to the global ``MultiScaleDeformableAttention`` (deformable_detr.py:392)."""

    @staticmethod
    def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
        lib = _lib.lib()
        B, S, M, D = value.shape
        L = spatial_shapes.shape[0]
        Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
        _chk(spatial_shapes, "spatial_shapes", torch.int64)
        _chk(level_start_index, "level_start_index", torch.int64)
        if value.dtype == torch.float64:  # AT_DISPATCH_FLOATING_TYPES (ms_deform_attn_cuda.cu:67): double is served too
            for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")):
                _chk(t, n, torch.float64)
            out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
            st = lib.egtr_msda_forward_f64(_stream(), value.data_ptr(), spatial_shapes.data_ptr(),
                                           level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                           attn_weight.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr())
            _lib.check(st, "ms_deform_attn_forward")
            return out
        _chk(sampling_loc, "sampling_loc", torch.float32)
        _chk(attn_weight, "attn_weight", torch.float32)
        if value.dtype == torch.float32:
            _chk(value, "value")
            out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
            st = lib.egtr_msda_forward_f32(_stream(), value.data_ptr(), spatial_shapes.data_ptr(),
                                           level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                           attn_weight.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr())
        elif value.dtype == torch.bfloat16:
            _chk(value, "value")
            out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
            st = lib.egtr_msda_forward_bf16(_stream(), value.data_ptr(), spatial_shapes.data_ptr(),
                                            level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                            attn_weight.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr())
        else:
            raise RuntimeError(f"ms_deform_attn_forward: unsupported dtype {value.dtype}")
        _lib.check(st, "ms_deform_attn_forward")
        return out

    @staticmethod
    def ms_deform_attn_forward_fused(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits,
                                     reference_points, want_weights=False, keep_mask=None, value_bias=None, keep_bits=None):
        """Forward with the softmax over the L*P logits and ``loc = ref + offset / (W, H)`` (2-d reference points) or
        ``loc = box.xy + offset / P * box.wh * 0.5`` (4-d reference boxes) computed in the kernel
        (deformable_detr.py:1055-1081).  fp32, M = 8, D = 32, L*P = 16; no autograd.
        sampling_offsets [B,Lq,M,L,P,2] / attn_logits [B,Lq,M,L*P] may be column blocks of one wider Linear output
        (any row stride, unit inner strides); keep_mask [B,S] bool: padded tokens are skipped (== zeroed value rows).
        keep_bits [B, ceil(S/32)] int32: the same mask packed one bit per token (bit t%32 of word t/32; what
        ``ops.level_geometry`` returns as its fifth value) -- an EXPLICIT argument: the caller that owns the mask hands the
        packed copy down; without it the byte mask is read by the kernel.
        value_bias [M*D]: ``value`` is the bias-free value projection and the bias is applied inside the kernel (times the
        sum of the in-range, unpadded corner weights).
        Returns (out [B,Lq,M*D], attention weights [B,Lq,M,L,P] or None)."""
        lib = _lib.lib()
        B, S, M, D = value.shape
        L = spatial_shapes.shape[0]
        Lq, P = sampling_offsets.shape[1], sampling_offsets.shape[4]
        for t, n in ((value, "value"), (reference_points, "reference_points")):
            _chk(t, n, torch.float32)
        _chk(spatial_shapes, "spatial_shapes", torch.int64)
        _chk(level_start_index, "level_start_index", torch.int64)
        if tuple(reference_points.shape) not in ((B, Lq, L, 2), (B, Lq, L, 4)):
            raise RuntimeError(f"ms_deform_attn_forward_fused: reference_points must be [B, Lq, L, 2] or "
                               f"[B, Lq, L, 4], got {tuple(reference_points.shape)}")
        # 4-d reference boxes: loc = box.xy + offset / P * box.wh * 0.5 (deformable_detr.py:1074-1081)
        entry = (lib.egtr_msda_forward_fused_box_f32 if reference_points.shape[-1] == 4
                 else lib.egtr_msda_forward_fused_vbias_f32)

        def rows(t, width, name):  # [B, Lq, width...] -> row stride in floats (dense inner dims, uniform row stride)
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError(f"{name} must be a float32 CUDA/HIP tensor")
            t2 = t.reshape(B, Lq, width) if t.is_contiguous() else t.flatten(2)
            if t2.stride(2) != 1 or (B > 1 and t2.stride(0) != Lq * t2.stride(1)):
                t2 = t2.contiguous()
            return t2, t2.stride(1)

        off2, ld_off = rows(sampling_offsets, M * L * P * 2, "sampling_offsets")
        log2, ld_log = rows(attn_logits, M * L * P, "attn_logits")
        km = kbits = None
        if keep_bits is not None:
            kbits = _check_keep_bits(keep_bits, B, S)
        elif keep_mask is not None:
            km = keep_mask.reshape(B, S).contiguous()
            km = km.view(torch.uint8) if km.dtype == torch.bool else (km != 0).to(torch.uint8)
            _chk(km, "keep_mask")
        out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
        wts = torch.empty(B, Lq, M, L, P, dtype=value.dtype, device=value.device) if want_weights else None
        vb = None
        if value_bias is not None:
            vb = _chk(value_bias.detach().contiguous(), "value_bias", torch.float32)
            if vb.numel() != M * D:
                raise RuntimeError(f"value_bias must have {M * D} elements, got {vb.numel()}")
        st = entry(_stream(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                   off2.data_ptr(), log2.data_ptr(), reference_points.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr(),
                   wts.data_ptr() if want_weights else None, ld_off, ld_log,
                   km.data_ptr() if km is not None else None, kbits.data_ptr() if kbits is not None else None,
                   vb.data_ptr() if vb is not None else None)
        _lib.check(st, "ms_deform_attn_forward_fused")
        return out, wts

    @staticmethod
    def ms_deform_attn_forward_fused_bf16(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits,
                                          reference_points, keep_mask=None, keep_bits=None):
        """bf16 counterpart of ``ms_deform_attn_forward_fused`` (M = 8, D = 32, L*P = 16): every tensor bf16, softmax and
        sampling locations formed in fp32 inside the kernel.  Returns out [B, Lq, M*D] bf16."""
        lib = _lib.lib()
        B, S, M, D = value.shape
        L = spatial_shapes.shape[0]
        Lq, P = sampling_offsets.shape[1], sampling_offsets.shape[4]
        bf = torch.bfloat16
        _chk(value, "value", bf)
        _chk(spatial_shapes, "spatial_shapes", torch.int64)
        _chk(level_start_index, "level_start_index", torch.int64)
        ref = _chk(reference_points.to(bf).contiguous(), "reference_points", bf)
        if tuple(ref.shape) != (B, Lq, L, 2):
            raise RuntimeError(f"reference_points must be [B, Lq, L, 2], got {tuple(ref.shape)}")

        def rows(t, width, name):
            if not t.is_cuda or t.dtype != bf:
                raise RuntimeError(f"{name} must be a bfloat16 CUDA/HIP tensor")
            t2 = t.reshape(B, Lq, width) if t.is_contiguous() else t.flatten(2)
            if t2.stride(2) != 1 or (B > 1 and t2.stride(0) != Lq * t2.stride(1)) or t2.data_ptr() % 8:
                t2 = t2.contiguous()
            return t2, t2.stride(1)

        off2, ld_off = rows(sampling_offsets, M * L * P * 2, "sampling_offsets")
        log2, ld_log = rows(attn_logits, M * L * P, "attn_logits")
        kbits = None
        if keep_bits is not None:   # one bit per token, handed down explicitly by the owner of the mask (ops.level_geometry)
            kbits = _check_keep_bits(keep_bits, B, S)
        elif keep_mask is not None:  # packed here, per call (the kernel keeps the bits of its image in LDS)
            kbits = pack_keep_bits(keep_mask, B, S)
        out = torch.empty(B, Lq, M * D, dtype=bf, device=value.device)
        st = lib.egtr_msda_forward_fused_bf16(_stream(), value.data_ptr(), spatial_shapes.data_ptr(),
                                              level_start_index.data_ptr(), off2.data_ptr(), log2.data_ptr(),
                                              ref.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr(), ld_off, ld_log, None,
                                              kbits.data_ptr() if kbits is not None else None)
        _lib.check(st, "ms_deform_attn_forward_fused_bf16")
        return out

    @staticmethod
    def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                                im2col_step):
        lib = _lib.lib()
        B, S, M, D = value.shape
        L = spatial_shapes.shape[0]
        Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
        _chk(spatial_shapes, "spatial_shapes", torch.int64)
        _chk(level_start_index, "level_start_index", torch.int64)
        if value.dtype == torch.float64:      # ms_deform_attn_cuda.cu:137 dispatches double as well
            for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight"),
                         (grad_output, "grad_output")):
                _chk(t, n, torch.float64)
            grad_value = torch.zeros_like(value)
            grad_loc = torch.empty_like(sampling_loc)
            grad_attn = torch.empty_like(attn_weight)
            st = lib.egtr_msda_backward_f64(_stream(), grad_output.data_ptr(), value.data_ptr(),
                                            spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                            sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Lq, P,
                                            grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr())
            _lib.check(st, "ms_deform_attn_backward")
            return grad_value, grad_loc, grad_attn
        if value.dtype == torch.bfloat16:     # bf16 values / upstream gradient, fp32 geometry and gradients
            _chk(value, "value", torch.bfloat16)
            _chk(grad_output, "grad_output", torch.bfloat16)
            _chk(sampling_loc, "sampling_loc", torch.float32)
            _chk(attn_weight, "attn_weight", torch.float32)
            gv32 = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
            grad_loc = torch.empty_like(sampling_loc)
            grad_attn = torch.empty_like(attn_weight)
            # (0 for the model's shapes: the kernels read the bf16 operands directly; other shapes widen them into a workspace)
            nws = int(lib.egtr_msda_backward_bf16_workspace_floats(B, S, M, D, L, Lq, P))
            ws = torch.empty(nws, dtype=torch.float32, device=value.device) if nws else None
            st = lib.egtr_msda_backward_bf16(_stream(), grad_output.data_ptr(), value.data_ptr(),
                                             spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                             sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Lq, P,
                                             gv32.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                                             ws.data_ptr() if ws is not None else None)
            _lib.check(st, "ms_deform_attn_backward")
            return gv32.to(torch.bfloat16), grad_loc, grad_attn
        for t, n in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight"),
                     (grad_output, "grad_output")):
            _chk(t, n, torch.float32)
        # accumulated with atomics (reference: zeros_like, cu:124); cleared by the call itself -- encoder-shaped calls inside
        # the first kernel of the pair, no 51 MB fill launch (egtr_msda_backward_out_f32)
        grad_value = torch.empty_like(value)
        grad_loc = torch.empty_like(sampling_loc)
        grad_attn = torch.empty_like(attn_weight)
        st = lib.egtr_msda_backward_out_f32(_stream(), grad_output.data_ptr(), value.data_ptr(),
                                        spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                        sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Lq,
                                        P, grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr())
        _lib.check(st, "ms_deform_attn_backward")
        return grad_value, grad_loc, grad_attn


def load_hip_kernels():
    """Counterpart of load_cuda_kernels() (model/load_custom.py:23): returns the kernel module object.
    Raises if libegtr_hip.so is missing -- the caller must not swallow that."""
    _lib.lib()
    return _MultiScaleDeformableAttention


# the reference's name, so `from .load_custom import load_cuda_kernels` keeps working
load_cuda_kernels = load_hip_kernels
