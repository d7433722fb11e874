"""Autograd bridges between PyTorch-ROCm tensors and the HIP kernels of libegtr_hip.so.

Every op here enqueues on torch's current HIP stream through the C ABI (include/egtr_hip.h); none has a CPU or
eager-PyTorch fallback -- a missing library raises ``egtr_amd._lib.EgtrHipError``.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from .load_custom import _chk, _stream, load_hip_kernels

_MSDA = None


def _c16(t):
    """Contiguous AND 16-byte aligned (a contiguous view with an odd storage offset is copied): what the C entries'
    vector loads require; they answer EGTR_E_UNSUPPORTED otherwise."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _msda():
    global _MSDA
    if _MSDA is None:
        _MSDA = load_hip_kernels()
    return _MSDA


class MultiScaleDeformableAttentionFunction(Function):
    """Same contract as the reference's autograd Function (model/deformable_detr.py:402-455)."""

    @staticmethod
    def forward(context, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        context.im2col_step = im2col_step
        if value.dtype == torch.bfloat16:  # bf16 values, fp32 sampling geometry (csrc: msda_fwd_q32_bf16)
            sampling_locations, attention_weights = sampling_locations.float(), attention_weights.float()
        output = _msda().ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                                sampling_locations, attention_weights, im2col_step)
        context.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                  attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(context, grad_output):
        value, shapes, lsi, loc, attn = context.saved_tensors
        grad_value, grad_loc, grad_attn = _msda().ms_deform_attn_backward(
            value, shapes, lsi, loc, attn, grad_output.contiguous(), context.im2col_step)
        # bf16 values: loc / attn were promoted to fp32 in forward(); autograd casts their gradients back
        return grad_value, None, None, grad_loc, grad_attn, None


class MSDAGeometryFunction(Function):
    """(sampling_offsets [B, Lq, M*L*P*2], attention logits [B, Lq, M*L*P], reference_points [B, Lq, L, 2 | 4]) ->
    (sampling_locations [B, Lq, M, L, P, 2], attention_weights [B, Lq, M, L, P]) under autograd: softmax + location
    arithmetic of model/deformable_detr.py:1055-1073 in one pass per direction (csrc/msda_geom.hip).
    ``logits`` None: ``offsets`` is the output [B, Lq, 3*M*L*P] of ONE nn.Linear over the concatenated weights (offsets
    columns first); its gradient is then written as one buffer as well."""

    @staticmethod
    def forward(ctx, offsets, logits, reference_points, spatial_shapes, M, L, P):
        lib = _lib.lib()
        B, Lq = offsets.shape[:2]
        n_off = M * L * P * 2
        both = offsets.reshape(B * Lq, -1)
        both = both if both.stride(1) == 1 else both.contiguous()
        if logits is None:
            off, lg = both[:, :n_off], both[:, n_off:]
        else:
            off = both
            lg = logits.reshape(B * Lq, -1)
            lg = lg if lg.stride(1) == 1 else lg.contiguous()
        ref = _chk(reference_points.contiguous(), "reference_points", torch.float32)
        shp = _chk(spatial_shapes.contiguous(), "spatial_shapes", torch.int64)
        loc = torch.empty(B, Lq, M, L, P, 2, dtype=torch.float32, device=off.device)
        probs = torch.empty(B, Lq, M, L, P, dtype=torch.float32, device=off.device)
        st = lib.egtr_msda_geometry_forward_f32(_stream(), off.data_ptr(), off.stride(0), lg.data_ptr(), lg.stride(0),
                                                ref.data_ptr(), ref.shape[-1], shp.data_ptr(), loc.data_ptr(),
                                                probs.data_ptr(), B * Lq, M, L, P)
        _lib.check(st, "egtr_msda_geometry_forward_f32")
        ctx.save_for_backward(off, ref, shp, probs)
        ctx.dims = (M, L, P)
        ctx.shapes = (offsets.shape, logits.shape if logits is not None else None)
        return loc, probs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loc, g_probs):
        lib = _lib.lib()
        off, ref, shp, probs = ctx.saved_tensors
        M, L, P = ctx.dims
        rows = off.shape[0]
        n_off = M * L * P * 2
        g_loc = _chk(g_loc.contiguous(), "grad_locations", torch.float32)
        g_probs = _chk(g_probs.contiguous(), "grad_weights", torch.float32)
        g_off = torch.empty(ctx.shapes[0], dtype=torch.float32, device=off.device)
        if ctx.shapes[1] is None:
            g_lg, p_lg, ld_off, ld_lg = None, g_off.data_ptr() + 4 * n_off, g_off.shape[-1], g_off.shape[-1]
        else:
            g_lg = torch.empty(ctx.shapes[1], dtype=torch.float32, device=off.device)
            p_lg, ld_off, ld_lg = g_lg.data_ptr(), n_off, n_off // 2
        g_ref = torch.empty_like(ref) if ctx.needs_input_grad[2] else None
        st = lib.egtr_msda_geometry_backward_f32(_stream(), g_loc.data_ptr(), g_probs.data_ptr(), probs.data_ptr(),
                                                 off.data_ptr(), off.stride(0), ref.data_ptr(), ref.shape[-1],
                                                 shp.data_ptr(), g_off.data_ptr(), ld_off, p_lg, ld_lg,
                                                 g_ref.data_ptr() if g_ref is not None else None, rows, M, L, P)
        _lib.check(st, "egtr_msda_geometry_backward_f32")
        return g_off, g_lg, g_ref, None, None, None, None


def msda_geometry_supported(offsets, logits, reference_points, M, L, P):
    """Shapes / dtypes served by MSDAGeometryFunction (else the ATen composition)."""
    def rows_ok(t):   # what egtr_msda_geometry_*_f32 asks of a row-strided operand: 16-byte base, row stride % 4 floats
        return t.data_ptr() % 16 == 0 and (t.stride(-1) != 1 or t.stride(-2) % 4 == 0)

    return (MSDA_GEOMETRY and offsets.is_cuda and offsets.dtype == torch.float32 and logits.dtype == torch.float32
            and reference_points.dtype == torch.float32 and L == 4 and P == 4 and M == 8
            and reference_points.shape[-1] in (2, 4) and offsets.dim() == 3 and logits.dim() == 3
            and reference_points.dim() == 4 and reference_points.shape[2] == L and M * L * P * 2 % 4 == 0
            and rows_ok(offsets) and rows_ok(logits))


MSDA_GEOMETRY = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


def msda_fused_supported(num_heads, channels, num_levels, num_points):
    """Shapes served by egtr_msda_forward_fused_f32 (the wave-per-query kernel)."""
    return num_heads == 8 and channels == 32 and num_levels * num_points == 16 and num_levels <= 4 \
        and num_points % 2 == 0


def msda_forward_fused(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits, reference_points,
                       want_weights=False, keep_mask=None, value_bias=None, keep_bits=None):
    """MSDA forward with its softmax / sampling-location prologue fused in (no autograd: inference path).  With
    ``value_bias`` (fp32 only) ``value`` is the bias-free value projection and the kernel applies the bias.
    ``keep_bits``: the bit-packed copy of ``keep_mask`` (``level_geometry``'s fifth result), passed down explicitly."""
    from .load_custom import load_hip_kernels
    k = load_hip_kernels()
    if value.dtype == torch.bfloat16:
        if want_weights or value_bias is not None:
            raise NotImplementedError("the bf16 fused MSDA forward returns no attention weights and takes no value_bias")
        return k.ms_deform_attn_forward_fused_bf16(value, spatial_shapes, level_start_index, sampling_offsets,
                                                   attn_logits, reference_points, keep_mask, keep_bits=keep_bits), None
    return k.ms_deform_attn_forward_fused(value, spatial_shapes, level_start_index, sampling_offsets, attn_logits,
                                          reference_points, want_weights, keep_mask, value_bias=value_bias,
                                          keep_bits=keep_bits)


class DecoderSelfAttentionFunction(Function):
    """softmax(q k^T) v per head + the retained [B, M, N, D] maps of scaled q and k
    (replaces model/deformable_detr.py:1170-1253; q must already carry the D^-1/2 scaling of :1166)."""

    @staticmethod
    def forward(ctx, q, k, v, num_heads, want_maps):
        lib = _lib.lib()
        B, N, MD = q.shape
        D = MD // num_heads
        for t, n in ((q, "q"), (k, "k"), (v, "v")):
            _chk(t, n, torch.float32)
        out = torch.empty_like(q)
        need_bwd = q.requires_grad or k.requires_grad or v.requires_grad
        qh = torch.empty(B, num_heads, N, D, dtype=q.dtype, device=q.device) if want_maps else None
        kh = torch.empty_like(qh) if want_maps else None
        lse = torch.empty(B, num_heads, N, dtype=q.dtype, device=q.device) if need_bwd else None
        st = lib.egtr_self_attn_forward_f32(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), B, N, num_heads, D,
                                            out.data_ptr(), qh.data_ptr() if want_maps else None,
                                            kh.data_ptr() if want_maps else None,
                                            lse.data_ptr() if need_bwd else None)
        _lib.check(st, "egtr_self_attn_forward_f32")
        ctx.num_heads = num_heads
        ctx.want_maps = want_maps
        if need_bwd:
            ctx.save_for_backward(q, k, v, out, lse)
        if want_maps:
            return out, qh, kh
        return out, None, None

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, grad_qh, grad_kh):
        lib = _lib.lib()
        q, k, v, out, lse = ctx.saved_tensors
        B, N, MD = q.shape
        M = ctx.num_heads
        grad_out = grad_out.contiguous()
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        st = lib.egtr_self_attn_backward_f32(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                             lse.data_ptr(), grad_out.data_ptr(), B, N, M, MD // M, gq.data_ptr(),
                                             gk.data_ptr(), gv.data_ptr())
        _lib.check(st, "egtr_self_attn_backward_f32")
        # the retained maps are pure re-layouts of q and k: their gradients fold straight back
        if grad_qh is not None:
            gq = gq + grad_qh.transpose(1, 2).reshape(B, N, MD)
        if grad_kh is not None:
            gk = gk + grad_kh.transpose(1, 2).reshape(B, N, MD)
        return gq, gk, gv, None, None


def decoder_self_attention(q, k, v, num_heads, want_maps=True):
    if (q.dtype == torch.bfloat16 and q.is_cuda and k.dtype == v.dtype == torch.bfloat16 and q.shape[-1] == 32 * num_heads
            and q.shape[1] <= 640 and not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad))):
        # bf16 model at inference: the kernel reads and writes bf16 itself (fp32 arithmetic) -- no cast launches around it
        lib = _lib.lib()
        B, N, MD = q.shape
        q2, k2, v2 = (t if (t.is_contiguous() and t.data_ptr() % 16 == 0) else t.contiguous() for t in (q, k, v))
        out = torch.empty_like(q2)
        qh = torch.empty(B, num_heads, N, 32, dtype=q.dtype, device=q.device) if want_maps else None
        kh = torch.empty_like(qh) if want_maps else None
        _lib.check(lib.egtr_self_attn_forward_bf16(_stream(), q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), B, N, num_heads, 32,
                                                   out.data_ptr(), qh.data_ptr() if want_maps else None,
                                                   kh.data_ptr() if want_maps else None), "egtr_self_attn_forward_bf16")
        return out, qh, kh
    if q.dtype != torch.float32:  # fp16 models / bf16 under autograd: the kernel computes in fp32, results go back to the model dtype
        o, qm, km = DecoderSelfAttentionFunction.apply(q.float().contiguous(), k.float().contiguous(),
                                                       v.float().contiguous(), num_heads, want_maps)
        return o.to(q.dtype), (qm.to(q.dtype) if qm is not None else None), (km.to(q.dtype) if km is not None else None)
    return DecoderSelfAttentionFunction.apply(q.contiguous(), k.contiguous(), v.contiguous(), num_heads, want_maps)


SKINNY_BACKWARD_FUSED = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6
# token-sized linears under autograd through TokenLinearFunction (split-bf16 forward / data / weight gradients); "0": plain autograd
TOKEN_LINEAR = os.environ.get("EGTR_TOKEN_LINEAR", "1") != "0"
SKINNY_MAX_ROWS = 4096  # above this the vendor GEMM (rocBLAS / hipBLASLt) fills the chip and is the right tool


class SkinnyLinearFunction(Function):
    """act((x W^T + b) * alpha) through egtr_linear_f32 (csrc/linear.hip).  Backward: egtr_linear_backward_f32 (data, weight
    and bias gradient in one launch) for N % 64 == 0, else vendor GEMMs + egtr_column_sum_f32."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, relu):
        lib = _lib.lib()
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = _chk(x.reshape(-1, K).contiguous(), "x", torch.float32)
        w = _chk(weight.contiguous(), "weight", torch.float32)
        b = _chk(bias.contiguous(), "bias", torch.float32) if bias is not None else None
        y = torch.empty(x2.shape[0], N, dtype=torch.float32, device=x.device)
        st = lib.egtr_linear_f32(_stream(), x2.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                 y.data_ptr(), x2.shape[0], K, N, float(alpha), 1 if relu else 0)
        _lib.check(st, "egtr_linear_f32")
        ctx.alpha, ctx.relu, ctx.has_bias = float(alpha), bool(relu), bias is not None
        if x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad):
            ctx.save_for_backward(x2, w, y if relu else None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        x2, w, y = ctx.saved_tensors
        g = grad_y.reshape(-1, grad_y.shape[-1])
        gb = None
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        N, K = w.shape
        if SKINNY_BACKWARD_FUSED and N % 64 == 0 and K % 64 == 0 and g.dtype == torch.float32:
            # gx, gw, gb (+ ReLU mask and alpha) in one launch (egtr_linear_backward_f32)
            lib = _lib.lib()
            g = _chk(_c16(g), "grad", torch.float32)
            M = g.shape[0]
            gx = torch.empty(M, K, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
            gw = torch.empty(N, K, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
            gb = torch.empty(N, dtype=torch.float32, device=g.device) if want_gb else None
            if gx is not None or gw is not None or gb is not None:
                st = lib.egtr_linear_backward_f32(_stream(), g.data_ptr(), y.data_ptr() if ctx.relu else None,
                                                  x2.data_ptr(), w.data_ptr(), ctx.alpha,
                                                  gx.data_ptr() if gx is not None else None,
                                                  gw.data_ptr() if gw is not None else None,
                                                  gb.data_ptr() if gb is not None else None, M, K, N)
                _lib.check(st, "egtr_linear_backward_f32")
            return (gx.view(*grad_y.shape[:-1], K) if gx is not None else None), gw, gb, None, None
        if ctx.relu and ctx.alpha == 1.0 and want_gb:
            g, gb = column_sum(g, relu_output=y)   # ReLU mask and bias gradient in one pass
        else:
            if ctx.relu:
                g = g * (y > 0).to(g.dtype)
            if ctx.alpha != 1.0:
                g = g * ctx.alpha
            if want_gb:
                gb = column_sum(g)
        gx = (g @ w).view(*grad_y.shape[:-1], w.shape[1]) if ctx.needs_input_grad[0] else None
        gw = g.t() @ x2 if ctx.needs_input_grad[1] else None
        return gx, gw, gb, None, None


def linear(x, weight, bias=None, alpha=1.0, relu=False):
    """nn.Linear (+ optional scale and ReLU).  Object-query-sized inputs on the GPU (rows <= SKINNY_MAX_ROWS,
    K % 64 == 0, fp32) run the hand-written skinny MFMA kernel (forward and backward); token-sized inputs (encoder,
    S ~ 12.5k rows per image) run the split-bf16 GEMM kernels -- in inference through ``module_linear`` /
    ``linear_split_bf16``, under autograd through ``TokenLinearFunction`` -- and the vendor GEMM where those do not apply
    (feature counts that are not multiples of 128 / 32, bf16 models).  (On CPU tensors -- host-logic tests -- F.linear.)"""
    rows = x.numel() // x.shape[-1]
    if x.is_cuda and x.dtype == torch.float32 and rows <= SKINNY_MAX_ROWS and x.shape[-1] % 64 == 0:
        return SkinnyLinearFunction.apply(x, weight, bias, alpha, relu)
    if (LINEAR_BF16 and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and rows <= LINEAR_BF16_MAX_ROWS and x.shape[-1] % 16 == 0 and weight.shape[0] % 32 == 0
            and (bias is None or bias.dtype == torch.bfloat16)
            and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad))):
        return linear_bf16(x, weight, bias, relu, alpha)
    if (relu and alpha == 1.0 and bias is not None and x.is_cuda and not torch.is_grad_enabled()
            and hasattr(torch, "_addmm_activation")):
        # token-sized GEMM with the ReLU in the hipBLASLt epilogue (saves one pass over the [S, 1024] activation)
        y = torch._addmm_activation(bias, x.reshape(-1, x.shape[-1]), weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], weight.shape[0])
    if (alpha == 1.0 and bias is not None and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and rows > SKINNY_MAX_ROWS and torch.is_grad_enabled() and TOKEN_LINEAR
            and (x.requires_grad or weight.requires_grad or bias.requires_grad)):
        return TokenLinearFunction.apply(x, weight, bias, relu)   # training, token-sized: bias gradient in one HIP pass
    y = torch.nn.functional.linear(x, weight, bias)
    if alpha != 1.0:
        y = y * alpha
    return torch.relu(y) if relu else y


# bf16 models: the encoder layer's feed-forward block + residual + LayerNorm (+ position rows) in one launch
# (csrc/ffn_bf16.hip).  "0": two vendor GEMMs + the LayerNorm launch.
FFN_BF16_FUSED = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


def ffn_bf16_supported(x, fc1, fc2, ln):
    return (FFN_BF16_FUSED and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and x.shape[-1] == 256
            and all(t.dtype == torch.bfloat16 for t in (fc1.weight, fc2.weight, ln.weight))
            and fc1.bias is not None and fc2.bias is not None and tuple(fc1.weight.shape)[1] == 256
            and tuple(fc2.weight.shape) == (256, fc1.weight.shape[0]) and fc1.weight.shape[0] % 32 == 0
            and fc1.weight.shape[0] <= 1024 and tuple(ln.weight.shape) == (256,))


def ffn_layernorm_bf16(x, fc1, fc2, ln, pos=None):
    """LayerNorm(x + fc2(relu(fc1(x)))) for a bf16 model in one launch (egtr_ffn_layernorm_bf16); with ``pos`` ([rows_p, 256]
    bf16, tiled over the rows) also returns the bf16 sum of the result and the position rows.  Inference only."""
    lib = _lib.lib()
    x2 = _chk(x.reshape(-1, 256).contiguous(), "x", torch.bfloat16)
    F_ = fc1.weight.shape[0]

    def pack():
        w1 = _chk(fc1.weight.detach().contiguous(), "fc1.weight", torch.bfloat16)
        w2 = _chk(fc2.weight.detach().contiguous(), "fc2.weight", torch.bfloat16)
        out = torch.empty(int(lib.egtr_ffn_packed_weights_bytes(F_)) // 2, dtype=torch.bfloat16, device=w1.device)
        _lib.check(lib.egtr_ffn_pack_weights_bf16(_stream(), w1.data_ptr(), w2.data_ptr(), 256, F_, out.data_ptr()),
                   "egtr_ffn_pack_weights_bf16")
        return out

    wpk = cached_weights(fc1, "ffn_bf16_packed", [fc1.weight, fc2.weight], pack)
    ts = [wpk] + [_chk(t.detach().contiguous(), n, torch.bfloat16)
                  for t, n in ((fc1.bias, "fc1.bias"), (fc2.bias, "fc2.bias"), (ln.weight, "ln.weight"), (ln.bias, "ln.bias"))]
    M = x2.shape[0]
    y = torch.empty_like(x2)
    yp = p2 = None
    prow = 1
    if pos is not None:
        p2 = _chk(pos.reshape(-1, 256).contiguous(), "pos", torch.bfloat16)
        prow = p2.shape[0]
        if M % prow != 0:
            raise ValueError("ffn_layernorm_bf16: pos must tile the rows")
        yp = torch.empty_like(x2)
    st = lib.egtr_ffn_layernorm_bf16(_stream(), x2.data_ptr(), *[t.data_ptr() for t in ts], float(ln.eps),
                                     p2.data_ptr() if p2 is not None else None, prow, y.data_ptr(),
                                     yp.data_ptr() if yp is not None else None, M, 256, fc1.weight.shape[0])
    _lib.check(st, "egtr_ffn_layernorm_bf16")
    y = y.view(x.shape)
    return y if pos is None else (y, yp.view(x.shape))


# bf16 models, object-query-sized rows (the decoder of the stress configuration: 4800 rows): csrc/linear_bf16.hip instead of the
# vendor library, whose choice for these shapes takes 20 us per layer.  "0": F.linear.
LINEAR_BF16 = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6
LINEAR_BF16_MAX_ROWS = 16384


def linear_bf16(x, weight, bias=None, relu=False, alpha=1.0):
    """act(alpha (x . W^T + b)) for bf16 tensors (egtr_linear_bf16): fp32 accumulation, one rounding of the result.
    Inference only."""
    lib = _lib.lib()
    K, N = x.shape[-1], weight.shape[0]
    x2 = _chk(x.reshape(-1, K).contiguous(), "x", torch.bfloat16)
    w = _chk(weight.detach().contiguous(), "weight", torch.bfloat16)
    b = _chk(bias.detach().contiguous(), "bias", torch.bfloat16) if bias is not None else None
    if weight.shape[1] != K:
        raise ValueError("linear_bf16: weight must be [N, K]")
    M = x2.shape[0]
    if (x2.data_ptr() | w.data_ptr()) % 16 != 0:   # a view at an odd offset: the kernel's 16-byte operand loads need alignment
        y = torch.nn.functional.linear(x2, w, b)
        if alpha != 1.0:
            y = y * alpha
        return (torch.relu(y) if relu else y).view(*x.shape[:-1], N)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    if M > 0:
        st = lib.egtr_linear_bf16(_stream(), x2.data_ptr(), K, w.data_ptr(), b.data_ptr() if b is not None else None,
                                  y.data_ptr(), N, M, N, K, 1 if relu else 0, float(alpha))
        _lib.check(st, "egtr_linear_bf16")
    return y.view(*x.shape[:-1], N)


def column_sum(g, relu_output=None, inplace=False):
    """g [M, N] fp32 -> column sums [N] (the bias gradient of a linear layer), egtr_column_sum_f32: one launch for
    object-query-sized M.  With ``relu_output`` (the layer's post-ReLU output) returns (g * [y > 0], its column sums);
    ``inplace``: the masked gradient overwrites ``g`` (every element is read and written by the same thread)."""
    lib = _lib.lib()
    g = _chk(_c16(g), "grad", torch.float32)
    M, N = g.shape
    ws = torch.empty(int(lib.egtr_column_sum_workspace_floats(M, N)), dtype=torch.float32, device=g.device)
    out = torch.empty(N, dtype=torch.float32, device=g.device)
    gm = (g if inplace else torch.empty_like(g)) if relu_output is not None else None
    _lib.check(lib.egtr_column_sum_f32(_stream(), g.data_ptr(),
                                       _chk(_c16(relu_output), "relu_output", torch.float32).data_ptr() if gm is not None else None,
                                       gm.data_ptr() if gm is not None else None, ws.data_ptr(), out.data_ptr(), M, N),
               "egtr_column_sum_f32")
    return out if gm is None else (gm, out)


def weighted_column_sum(g, row_weight):
    """sum_r row_weight[r] * g[r, :] for g [M, N] fp32 (egtr_weighted_column_sum_f32)."""
    lib = _lib.lib()
    g = _chk(g.contiguous(), "g", torch.float32)
    w = _chk(row_weight.reshape(-1).contiguous(), "row_weight", torch.float32)
    M, N = g.shape
    if w.numel() != M:
        raise RuntimeError("weighted_column_sum: one weight per row expected")
    ws = torch.empty(int(lib.egtr_column_sum_workspace_floats(M, N)), dtype=torch.float32, device=g.device)
    out = torch.empty(N, dtype=torch.float32, device=g.device)
    _lib.check(lib.egtr_weighted_column_sum_f32(_stream(), g.data_ptr(), w.data_ptr(), ws.data_ptr(), out.data_ptr(), M, N),
               "egtr_weighted_column_sum_f32")
    return out


class TokenLinearFunction(Function):
    """nn.Linear (+ ReLU) on token-sized inputs in TRAINING (reference: the encoder / cross-attention nn.Linear layers,
    model/deformable_detr.py:1049, 1053-1058, 1102, 1337-1343, under autograd).  Forward and data gradient g W run on the
    bf16 matrix cores through the exact three-way split (csrc/gemm_split.hip; W and W^T are re-tiled by one launch each,
    egtr_gemm_split_tile_weights_f32) where the shape allows (out features % 128, reduction % 32), else on the vendor
    GEMM; the weight gradient x^T g is the vendor GEMM autograd would call; the bias gradient -- a [rows, N] column sum
    per layer, with the ReLU mask applied on the way -- is egtr_column_sum_f32 (one pass instead of threshold_backward +
    a generic reduction).  EGTR_TOKEN_LINEAR=0 keeps plain autograd; EGTR_GEMM_SPLIT_BF16=0 keeps the vendor GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2 = x.reshape(-1, x.shape[-1])
        N, K = weight.shape
        ctx.split = GEMM_SPLIT_BF16 and x2.shape[0] >= GEMM_SPLIT_MIN_ROWS and weight.stride(1) == 1
        ctx.wt_t = None
        if ctx.split and N % 128 == 0 and K % 128 == 0 and ctx.needs_input_grad[0]:
            wt, ctx.wt_t = gemm_split_tile_pair(weight)   # W^T for the data gradient, same launch
            y = linear_split_bf16(x2, wt, bias, N, relu=relu)
        elif ctx.split and N % 128 == 0 and K % 32 == 0:
            y = linear_split_bf16(x2, gemm_split_tile(weight), bias, N, relu=relu)
        elif relu and hasattr(torch, "_addmm_activation"):
            y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False)
        else:
            y = torch.addmm(bias, x2, weight.t())
            if relu:
                y = torch.relu_(y)
        ctx.relu = bool(relu)
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.in_shape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        x2, weight, y = ctx.saved_tensors
        N = weight.shape[0]
        g = grad_y.reshape(-1, N)
        g = _chk(g if g.is_contiguous() else g.contiguous(), "grad", torch.float32)
        if ctx.relu:
            g, gb = column_sum(g, relu_output=y)
        else:
            gb = column_sum(g)
        gx = None
        if ctx.needs_input_grad[0]:
            K = weight.shape[1]
            if ctx.split and K % 128 == 0 and N % 32 == 0:   # g W = "linear" with W^T
                wt_t = ctx.wt_t if ctx.wt_t is not None else gemm_split_tile(weight, transposed=True)
                gx = linear_split_bf16(g, wt_t, None, K)
            else:
                gx = g.mm(weight)
            gx = gx.view(ctx.in_shape)
        gw = None
        if ctx.needs_input_grad[1]:
            K = weight.shape[1]
            if ctx.split and GEMM_SPLIT_WGRAD and N % 128 == 0 and K % 128 == 0 and x2.stride(1) == 1:
                gw = linear_split_bf16_wgrad(g, x2)
            else:
                # the product autograd forms for addmm (x^T g, viewed transposed): the same vendor kernel as the plain path
                gw = x2.t().mm(g).t()
        return gx, gw, gb if ctx.needs_input_grad[2] else None, None


# ---- training form of the encoder layer (round 4): ONE autograd Function per layer -------------------------------------------
# "0": the per-op composition of rounds 2 / 3 (TokenLinearFunction + F.dropout + AddLayerNormFunction + clamp_nonfinite_ ...)
ENCODER_TRAIN_FUSED = os.environ.get("EGTR_ENCODER_TRAIN_FUSED", "1") != "0"
# training forward of the relation head on the split-bf16 arithmetic (rel_head_fwd_x6 with the activation stores); 0: exact-f32 kernel
REL_HEAD_TRAIN_X6 = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


def _host_array(ctype, vals):
    import ctypes
    return (ctype * len(vals))(*vals)


def linear_split_ex(problems, M, K):
    """Up to 8 token-sized linears with the same M and K in one launch of the split-bf16 GEMM with the training step's epilogue
    options (egtr_linear_split_bf16_ex_f32).  ``problems``: dicts with x [M, >=K] (unit inner stride), wt (tiled weight), N,
    and optionally b, relu, out ([M, N] view with unit inner stride), pos ([pos_rows, K]), row_keep ([M] uint8), relu_ref
    ([M, N]), add1 / add2 ([M, N], may alias out), colpart ([ceil(M / 32), N]).  Returns the outputs.  No autograd."""
    import ctypes
    lib = _lib.lib()
    n = len(problems)
    P, I = ctypes.c_void_p, ctypes.c_int
    outs = []
    for it in problems:
        y = it.get("out")
        if y is None:
            y = torch.empty(M, int(it["N"]), dtype=torch.float32, device=it["x"].device)
        outs.append(y)

    def ptrs(key):
        vals = [(it.get(key).data_ptr() if it.get(key) is not None else None) for it in problems]
        return _host_array(P, vals) if any(v is not None for v in vals) else None

    def ld(key):
        return _host_array(I, [(it[key].stride(0) if it.get(key) is not None else 0) for it in problems])

    for it in problems:
        for key in ("x", "relu_ref", "add1", "add2"):
            t = it.get(key)
            if t is not None and (t.stride(-1) != 1 or t.dtype != torch.float32 or not t.is_cuda):
                raise RuntimeError(f"linear_split_ex: {key} must be a float32 device tensor with unit inner stride")
        if it.get("add1") is not None and it.get("add2") is not None and it["add1"].stride(0) != it["add2"].stride(0):
            raise RuntimeError("linear_split_ex: add1 and add2 must share their row stride")
    ldadd = _host_array(I, [((it.get("add1") if it.get("add1") is not None else it.get("add2")).stride(0)
                             if (it.get("add1") is not None or it.get("add2") is not None) else 0) for it in problems])
    st = lib.egtr_linear_split_bf16_ex_f32(
        _stream(), n, _host_array(P, [it["x"].data_ptr() for it in problems]),
        _host_array(I, [it["x"].stride(0) for it in problems]), _host_array(P, [it["wt"].data_ptr() for it in problems]),
        _host_array(P, [(it["b"].data_ptr() if it.get("b") is not None else None) for it in problems]),
        _host_array(P, [y.data_ptr() for y in outs]), _host_array(I, [y.stride(0) for y in outs]),
        _host_array(I, [int(it["N"]) for it in problems]), _host_array(I, [1 if it.get("relu") else 0 for it in problems]),
        int(M), int(K), ptrs("pos"),
        _host_array(I, [(it["pos"].shape[0] if it.get("pos") is not None else 1) for it in problems]),
        ptrs("row_keep"), ptrs("relu_ref"), ld("relu_ref"), ptrs("add1"), ptrs("add2"), ldadd, ptrs("colpart"))
    _lib.check(st, "egtr_linear_split_bf16_ex_f32")
    return outs


def dropout_add_layernorm(x, residual, keep, scale, weight, bias, eps, flag=None):
    """LayerNorm(residual + keep * scale * x) over rows of 256 channels in one pass (egtr_dropout_add_layernorm_f32); ``keep``
    uint8 [rows, 256] or None; ``flag`` (int32 [1], optional) is OR-ed with 1 when an output element is non-finite."""
    lib = _lib.lib()
    rows = x.shape[0]
    y = torch.empty_like(x)
    st = lib.egtr_dropout_add_layernorm_f32(_stream(), x.data_ptr(), residual.data_ptr(),
                                            keep.data_ptr() if keep is not None else None, float(scale), weight.data_ptr(),
                                            bias.data_ptr(), y.data_ptr(), rows, 256, float(eps),
                                            flag.data_ptr() if flag is not None else None)
    _lib.check(st, "egtr_dropout_add_layernorm_f32")
    return y


def dropout_add_layernorm_backward(x, residual, keep, scale, weight, eps, grad_y, flag=None, y_out=None, clamp_value=0.0):
    """Backward of ``dropout_add_layernorm``: (grad_sum, grad_x (is grad_sum without dropout), [d gamma | d beta | d bias])."""
    lib = _lib.lib()
    rows = x.shape[0]
    gs = torch.empty_like(x)
    gx = torch.empty_like(x) if keep is not None else None
    ws = torch.empty(int(lib.egtr_dropout_add_layernorm_backward_workspace_floats(rows)), dtype=torch.float32, device=x.device)
    out = torch.empty(768, dtype=torch.float32, device=x.device)
    st = lib.egtr_dropout_add_layernorm_backward_f32(
        _stream(), x.data_ptr(), residual.data_ptr(), keep.data_ptr() if keep is not None else None, float(scale),
        weight.data_ptr(), grad_y.data_ptr(), flag.data_ptr() if flag is not None else None,
        y_out.data_ptr() if (flag is not None and y_out is not None) else None, float(clamp_value), gs.data_ptr(),
        gx.data_ptr() if gx is not None else None, ws.data_ptr(), out.data_ptr(), rows, 256, float(eps))
    _lib.check(st, "egtr_dropout_add_layernorm_backward_f32")
    return gs, (gx if gx is not None else gs), out


def _wgrad_ex(g, x, x_pos=None, row_keep=None):
    """g [M, N]^T . (x [+ x_pos rows]) [M, K] -> [N, K] with optional row mask on g (egtr_linear_split_bf16_wgrad_ex_f32)."""
    lib = _lib.lib()
    M, N = g.shape
    K = x.shape[1]
    ws = torch.empty(int(lib.egtr_linear_split_bf16_wgrad_workspace_floats(M, N, K)), dtype=torch.float32, device=g.device)
    gw = torch.empty(N, K, dtype=torch.float32, device=g.device)
    st = lib.egtr_linear_split_bf16_wgrad_ex_f32(
        _stream(), g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), gw.data_ptr(), ws.data_ptr(), M, N, K,
        x_pos.data_ptr() if x_pos is not None else None, x_pos.shape[0] if x_pos is not None else 1,
        row_keep.data_ptr() if row_keep is not None else None)
    _lib.check(st, "egtr_linear_split_bf16_wgrad_ex_f32")
    return gw


def _rows256(t):
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or t2.stride(0) != t2.shape[1] or t2.data_ptr() % 16:
        t2 = t2.contiguous()
        if t2.data_ptr() % 16:
            t2 = t2.clone()
    return t2


class EncoderLayerTrainFunction(Function):
    """One Deformable-DETR encoder layer in TRAINING as a single autograd node (reference: DeformableDetrEncoderLayer.forward
    in train mode, model/deformable_detr.py:1283-1358, with DeformableDetrMultiscaleDeformableAttention.forward, :1026-1104):

        value = mask(value_proj(x));  [offsets | logits] = Linear_cat(x + pos)  (pos added while the GEMM loads its operand)
        (loc, attn) = softmax / sampling locations;  ctx = MSDA(value, loc, attn)
        y1 = LayerNorm1(x + dropout(output_proj(ctx)));  y2 = LayerNorm2(y1 + dropout(fc2(relu(fc1(y1)))))
        y2 = clamp(y2) iff y2 holds an inf / nan (flag on the device, no host synchronisation)

    Why one node: the per-op composition of rounds 2 / 3 spent, per layer and step, ~25 ATen launches on glue around the
    same kernels -- gradient accumulation adds where branches meet, dropout forward / backward passes, `x + pos`, masked_fill and
    its backward, isfinite / clamp passes, threshold_backward + bias-gradient column sums over the [rows, 1024] activation
    (profiles/r04_train_gaps.txt: 5.5 ms of ATen elementwise kernels per step).  Here those are epilogue options of the split-bf16
    GEMMs (egtr_linear_split_bf16_ex_f32: row mask, ReLU backward, branch accumulation, bias-gradient partials) and of the two
    dropout + residual + LayerNorm kernels (csrc/enc_train.hip).  Dropout masks are bytes drawn by one bernoulli_ per layer
    (``masks`` hands in fixed ones for tests).  Arithmetic per product: exactly TokenLinearFunction's (six-term bf16 split)."""

    @staticmethod
    def forward(ctx, x, pos, ref, keep_rows, shapes, lsi, p_drop, masks, ln_eps, so_w, so_b, aw_w, aw_b, vp_w, vp_b, op_w,
                op_b, ln1_w, ln1_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_w, ln2_b):
        lib = _lib.lib()
        B, S, D = x.shape
        M = B * S
        dev = x.device
        x2 = _rows256(x.detach())
        pos2 = _rows256(pos.detach())
        rk = None
        if keep_rows is not None:
            rk = keep_rows.reshape(-1).contiguous()
            rk = rk.view(torch.uint8) if rk.dtype == torch.bool else rk.to(torch.uint8)
        bb = torch.cat([so_b.detach(), aw_b.detach()], 0)
        # every weight of the layer (and its transpose, for the data gradients) into the GEMM's operand stream: one launch;
        # sampling_offsets | attention_weights as ONE [384, 256] weight without a materialised concatenation
        (wt_v, wtT_v), (wt_b, wtT_b), (wt_o, wtT_o), (wt_1, wtT_1), (wt_2, wtT_2) = gemm_split_tile_pairs(
            [vp_w, (so_w, aw_w), op_w, fc1_w, fc2_w])
        F1 = fc1_w.shape[0]
        nb = so_w.shape[0] + aw_w.shape[0]
        n_off = so_w.shape[0]
        # value projection (padded rows zeroed in the epilogue) + offsets / logits projection of x + pos: one launch
        value, both = linear_split_ex([dict(x=x2, wt=wt_v, N=D, b=vp_b.detach(), row_keep=rk),
                                       dict(x=x2, wt=wt_b, N=nb, b=bb, pos=pos2)], M, D)
        Mh, L, P_ = 8, shapes.shape[0], n_off // (8 * shapes.shape[0] * 2)
        refc = _chk(ref.detach().contiguous(), "reference_points", torch.float32)
        shp = _chk(shapes.contiguous(), "spatial_shapes", torch.int64)
        loc = torch.empty(B, S, Mh, L, P_, 2, dtype=torch.float32, device=dev)
        attn = torch.empty(B, S, Mh, L, P_, dtype=torch.float32, device=dev)
        off, lg = both[:, :n_off], both[:, n_off:]
        _lib.check(lib.egtr_msda_geometry_forward_f32(_stream(), off.data_ptr(), off.stride(0), lg.data_ptr(), lg.stride(0),
                                                      refc.data_ptr(), refc.shape[-1], shp.data_ptr(), loc.data_ptr(),
                                                      attn.data_ptr(), M, Mh, L, P_), "egtr_msda_geometry_forward_f32")
        value4 = value.view(B, S, Mh, D // Mh)
        att = _msda().ms_deform_attn_forward(value4, shp, lsi, loc, attn, 64).view(M, D)
        a = linear_split_ex([dict(x=att, wt=wt_o, N=D, b=op_b.detach())], M, D)[0]
        p = float(p_drop)
        scale = 1.0 / (1.0 - p) if p > 0.0 else 1.0
        m1 = m2 = None
        if masks is not None:
            m1, m2 = masks
        elif p > 0.0:
            mm = torch.empty(2, M, D, dtype=torch.uint8, device=dev).bernoulli_(1.0 - p)
            m1, m2 = mm[0], mm[1]
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        eps1, eps2 = float(ln_eps[0]), float(ln_eps[1])
        y1 = dropout_add_layernorm(a, x2, m1, scale, ln1_w.detach(), ln1_b.detach(), eps1)
        h = linear_split_ex([dict(x=y1, wt=wt_1, N=F1, b=fc1_b.detach(), relu=True)], M, D)[0]
        f = linear_split_ex([dict(x=h, wt=wt_2, N=D, b=fc2_b.detach())], M, F1)[0]
        y2 = dropout_add_layernorm(f, y1, m2, scale, ln2_w.detach(), ln2_b.detach(), eps2, flag=flag)
        cv = torch.finfo(torch.float32).max - 1000
        _lib.check(lib.egtr_clamp_if_flag_f32(_stream(), y2.data_ptr(), None, y2.numel(), flag.data_ptr(), cv, 0),
                   "egtr_clamp_if_flag_f32")
        ctx.save_for_backward(x2, pos2, refc, shp, lsi, rk, value, both, loc, attn, att, a, m1, y1, h, f, m2, y2, flag,
                              wtT_v, wtT_b, wtT_o, wtT_1, wtT_2, ln1_w, ln2_w)
        ctx.dims = (B, S, D, F1, nb, n_off, Mh, L, P_, scale, cv, eps1, eps2)
        ctx.in_shape = x.shape
        ctx.pos_shape = pos.shape
        return y2.view(B, S, D)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_y2):
        lib = _lib.lib()
        (x2, pos2, refc, shp, lsi, rk, value, both, loc, attn, att, a, m1, y1, h, f, m2, y2, flag,
         wtT_v, wtT_b, wtT_o, wtT_1, wtT_2, ln1_w, ln2_w) = ctx.saved_tensors
        B, S, D, F1, nb, n_off, Mh, L, P_, scale, cv, eps1, eps2 = ctx.dims
        M = B * S
        dev = x2.device
        g = _rows256(g_y2)
        # LayerNorm2 + dropout backward (+ the clamp's gradient mask when the forward raised its flag)
        gs2, gf, gbb2 = dropout_add_layernorm_backward(f, y1, m2, scale, ln2_w.detach(), eps2, g, flag, y2, cv)
        d_w2 = linear_split_bf16_wgrad(gf, h)
        colp = torch.empty((M + 31) // 32, F1, dtype=torch.float32, device=dev)
        g_h = linear_split_ex([dict(x=gf, wt=wtT_2, N=F1, relu_ref=h, colpart=colp)], M, D)[0]   # ReLU mask + bias partials
        d_b1 = column_sum(colp)
        d_w1 = linear_split_bf16_wgrad(g_h, y1)
        linear_split_ex([dict(x=g_h, wt=wtT_1, N=D, add1=gs2, out=gs2)], M, F1)                  # gs2 <- d loss / d y1
        del g_h
        gs1, ga, gbb1 = dropout_add_layernorm_backward(a, x2, m1, scale, ln1_w.detach(), eps1, gs2)
        d_wo = linear_split_bf16_wgrad(ga, att)
        g_att = linear_split_ex([dict(x=ga, wt=wtT_o, N=D)], M, D)[0]
        g_value, g_loc, g_attn = _msda().ms_deform_attn_backward(value.view(B, S, Mh, D // Mh), shp, lsi, loc, attn,
                                                                 g_att.view(B, S, D), 64)
        g_both = torch.empty(M, nb, dtype=torch.float32, device=dev)
        off = both[:, :n_off]
        _lib.check(lib.egtr_msda_geometry_backward_f32(
            _stream(), g_loc.data_ptr(), g_attn.data_ptr(), attn.data_ptr(), off.data_ptr(), off.stride(0), refc.data_ptr(),
            refc.shape[-1], shp.data_ptr(), g_both.data_ptr(), nb, g_both.data_ptr() + 4 * n_off, nb, None, M, Mh, L, P_),
            "egtr_msda_geometry_backward_f32")
        d_bb = column_sum(g_both)
        d_wb = _wgrad_ex(g_both, x2, x_pos=pos2)
        g_qin = linear_split_ex([dict(x=g_both, wt=wtT_b, N=D)], M, nb)[0]                       # = d loss / d pos as well
        gv2 = g_value.view(M, D)
        d_bv = column_sum(gv2) if rk is None else weighted_column_sum(gv2, rk.to(torch.float32))
        d_wv = _wgrad_ex(gv2, x2, row_keep=rk)
        linear_split_ex([dict(x=gv2, wt=wtT_v, N=D, row_keep=rk, add1=gs1, add2=g_qin, out=gs1)], M, D)   # gs1 <- d loss / d x
        g_pos = g_qin.view(B, S, D)
        if tuple(ctx.pos_shape) != (B, S, D):
            g_pos = g_pos.sum_to_size(ctx.pos_shape)
        return (gs1.view(ctx.in_shape), g_pos, None, None, None, None, None, None, None,
                d_wb[:n_off], d_bb[:n_off], d_wb[n_off:], d_bb[n_off:], d_wv, d_bv, d_wo, gbb1[512:768],
                gbb1[0:256], gbb1[256:512], d_w1, d_b1, d_w2, gbb2[512:768], gbb2[0:256], gbb2[256:512])


class DecoderValueProjTrainFunction(Function):
    """The cross-attention value projections of ALL decoder layers in training as one autograd node (reference: value_proj +
    masked_fill of every DeformableDetrMultiscaleDeformableAttention of the decoder, model/deformable_detr.py:1048-1052):
    values_l = mask(enc W_l^T + b_l), l = 0 .. Ld - 1.  Forward: ONE grouped launch of the split-bf16 GEMM (the Ld products
    share the operand rows and the grid; padded rows zeroed in the epilogue).  Backward: d enc = sum_l mask (g_l W_l) as a
    chain of data-gradient products that accumulate into one buffer (no AccumulateGrad adds, no masked_fill backward), weight
    gradients with the row mask applied on load, bias gradients as mask-weighted column sums.  Returns Ld separate tensors
    (views of one stacked tensor would make autograd build a zero-filled [Ld, B, S, 256] gradient per layer)."""

    @staticmethod
    def forward(ctx, enc, keep_rows, *wb):
        nl = len(wb) // 2
        ws, bs = wb[:nl], wb[nl:]
        B, S, D = enc.shape
        M = B * S
        x2 = _rows256(enc.detach())
        rk = None
        if keep_rows is not None:
            rk = keep_rows.reshape(-1).contiguous()
            rk = rk.view(torch.uint8) if rk.dtype == torch.bool else rk.to(torch.uint8)
        tiles = []
        for i0 in range(0, nl, 8):
            tiles += gemm_split_tile_pairs(list(ws[i0:i0 + 8]))
        outs = []
        for i0 in range(0, nl, 8):
            outs += linear_split_ex([dict(x=x2, wt=tiles[i][0], N=D, b=bs[i].detach(), row_keep=rk)
                                     for i in range(i0, min(nl, i0 + 8))], M, D)
        ctx.save_for_backward(x2, rk, *[t[1] for t in tiles])
        ctx.set_materialize_grads(False)   # an unused layer's values: None, not a zero tensor to multiply out
        ctx.nl = nl
        ctx.in_shape = enc.shape
        return tuple(o.view(B, S, D) for o in outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        x2, rk = ctx.saved_tensors[:2]
        wtT = ctx.saved_tensors[2:]
        nl = ctx.nl
        M, D = x2.shape
        g_enc = None
        d_w, d_b = [], []
        rkf = rk.to(torch.float32) if rk is not None else None
        for i in range(nl):
            if gs[i] is None:
                d_w.append(None)
                d_b.append(None)
                continue
            g = _rows256(gs[i])
            d_b.append(column_sum(g) if rkf is None else weighted_column_sum(g, rkf))
            d_w.append(_wgrad_ex(g, x2, row_keep=rk))
            if g_enc is None:
                g_enc = linear_split_ex([dict(x=g, wt=wtT[i], N=D, row_keep=rk)], M, D)[0]
            else:
                linear_split_ex([dict(x=g, wt=wtT[i], N=D, row_keep=rk, add1=g_enc, out=g_enc)], M, D)
        return (g_enc.view(ctx.in_shape) if g_enc is not None else None, None, *d_w, *d_b)


def decoder_values_train_supported(enc, attention_mask, layers):
    eligible = (ENCODER_TRAIN_FUSED and GEMM_SPLIT_BF16 and torch.is_grad_enabled() and torch.is_tensor(enc) and enc.is_cuda
                and enc.dtype == torch.float32 and enc.dim() == 3 and enc.shape[0] * enc.shape[1] > SKINNY_MAX_ROWS)
    ok = (eligible and enc.shape[-1] == 256
          and all(tuple(l.encoder_attn.value_proj.weight.shape) == (256, 256) and l.encoder_attn.value_proj.bias is not None
                  and l.encoder_attn.value_proj.weight.dtype == torch.float32 for l in layers)
          and (attention_mask is None or tuple(attention_mask.shape) == tuple(enc.shape[:2])))
    return _gate("decoder_values_train", eligible, ok, lambda: f"encoder states {tuple(enc.shape)}: 256 channels, 256 -> 256 value "
                                                               "projections with biases served")


def decoder_values_train(enc, attention_mask, layers):
    """[value_proj_l(enc) with padded rows zeroed for l in layers] -- see DecoderValueProjTrainFunction."""
    return DecoderValueProjTrainFunction.apply(
        enc, attention_mask, *[l.encoder_attn.value_proj.weight for l in layers],
        *[l.encoder_attn.value_proj.bias for l in layers])


class DropoutAddLayerNormFunction(Function):
    """LayerNorm(residual + dropout(x)) over 256 channels as one pass per direction (csrc/enc_train.hip) -- the decoder layer's
    three "dropout, add, LayerNorm" steps in training (model/deformable_detr.py:1436-1438, 1455-1457, 1466-1468): instead of
    fused_dropout + add_layernorm forward and layer-norm backward + masked_scale backward.  The mask is a byte tensor drawn by
    bernoulli_ (``keep`` hands in a fixed one for tests)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps, p, keep):
        x2, r2 = _rows256(x.detach()), _rows256(residual.detach())
        scale = 1.0
        if keep is None and p > 0.0:
            keep = torch.empty(x2.shape, dtype=torch.uint8, device=x2.device).bernoulli_(1.0 - p)
        if keep is not None:
            scale = 1.0 / (1.0 - p)
        y = dropout_add_layernorm(x2, r2, keep, scale, weight.detach(), bias.detach(), eps)
        ctx.save_for_backward(x2, r2, keep, weight)
        ctx.cfg = (float(eps), float(scale))
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, r2, keep, weight = ctx.saved_tensors
        eps, scale = ctx.cfg
        gs, gx, gbb = dropout_add_layernorm_backward(x2, r2, keep, scale, weight.detach(), eps, _rows256(gy))
        return gx.view(gy.shape), gs.view(gy.shape), gbb[0:256], gbb[256:512], None, None, None


def dropout_add_layer_norm(x, residual, ln, p, training, keep=None):
    """ln(residual + dropout(x, p, training)); fp32 device tensors of 256 channels under autograd take the one-pass kernels."""
    if (ENCODER_TRAIN_FUSED and training and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
            and x.shape[-1] == 256 and residual.shape == x.shape and 0.0 <= p < 1.0):
        return DropoutAddLayerNormFunction.apply(x, residual, ln.weight, ln.bias, ln.eps, float(p), keep)
    return add_layer_norm(torch.nn.functional.dropout(x, p=p, training=training), residual, ln)


def encoder_layer_train_supported(layer, x, pos, ref, attention_mask, output_attentions):
    """The fused training node serves the reference's training configuration: fp32 on the GPU, token-sized rows, d_model 256,
    8 heads x 4 levels x 4 points, 2-d reference points, ReLU FFN with a hidden width that tiles (multiple of 128),
    activation_dropout 0 (the reference default), no attention maps requested."""
    sa = layer.self_attn
    eligible = (ENCODER_TRAIN_FUSED and GEMM_SPLIT_BF16 and torch.is_grad_enabled() and layer.training and not output_attentions
                and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
                and x.shape[0] * x.shape[1] > SKINNY_MAX_ROWS)
    ok = (eligible and x.shape[-1] == 256 and pos is not None and pos.dtype == torch.float32
          and tuple(pos.shape) == tuple(x.shape) and ref is not None and ref.dim() == 4 and ref.shape[-1] == 2
          and sa.n_heads == 8 and sa.n_levels == 4 and sa.n_points == 4 and ref.shape[2] == 4
          and layer.activation_fn is torch.nn.functional.relu and layer.activation_dropout == 0.0
          and layer.fc1.weight.shape[0] % 128 == 0 and tuple(layer.fc2.weight.shape) == (256, layer.fc1.weight.shape[0])
          and layer.fc1.weight.dtype == torch.float32 and 0.0 <= layer.dropout < 1.0
          and (attention_mask is None or tuple(attention_mask.shape) == tuple(x.shape[:2])))
    return _gate("encoder_layer_train", eligible, ok,
                 lambda: f"states {tuple(x.shape)}, {sa.n_heads} heads x {sa.n_levels} levels x {sa.n_points} points, fc1 "
                         f"{tuple(layer.fc1.weight.shape)}: the training node serves d_model 256, 8 x 4 x 4, ReLU, a hidden width "
                         "that is a multiple of 128, activation_dropout 0, 2-d reference points")


def encoder_layer_train(layer, x, attention_mask, pos, ref, spatial_shapes, level_start_index, masks=None):
    sa = layer.self_attn
    return EncoderLayerTrainFunction.apply(
        x, pos, ref, attention_mask, spatial_shapes, level_start_index, layer.dropout, masks,
        (layer.self_attn_layer_norm.eps, layer.final_layer_norm.eps), sa.sampling_offsets.weight, sa.sampling_offsets.bias, sa.attention_weights.weight, sa.attention_weights.bias,
        sa.value_proj.weight, sa.value_proj.bias, sa.output_proj.weight, sa.output_proj.bias,
        layer.self_attn_layer_norm.weight, layer.self_attn_layer_norm.bias, layer.fc1.weight, layer.fc1.bias,
        layer.fc2.weight, layer.fc2.bias, layer.final_layer_norm.weight, layer.final_layer_norm.bias)


# ---- decoder layer as ONE autograd node (training) ---------------------------------------------------------------------------
DECODER_TRAIN_FUSED = True   # module attribute, not an environment switch: tests patch it for the switch-off twin


def _skinny_fwd(x2, w, b, alpha=1.0, relu=False):
    """act((x W^T + b) * alpha) for object-query rows (egtr_linear_f32), plain tensors, no autograd."""
    lib = _lib.lib()
    M, K = x2.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
    _lib.check(lib.egtr_linear_f32(_stream(), x2.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                   y.data_ptr(), M, K, N, float(alpha), 1 if relu else 0), "egtr_linear_f32")
    return y


def _skinny_bwd(g, x2, w, alpha=1.0, relu_out=None, want_gb=True, add1=None, add2=None, out=None):
    """(grad_x [+ add1 + add2], grad_w, grad_b) of ``_skinny_fwd`` in one launch (egtr_linear_backward_acc_f32)."""
    lib = _lib.lib()
    M, N = g.shape
    K = w.shape[1]
    gx = out if out is not None else torch.empty(M, K, dtype=torch.float32, device=g.device)
    gw = torch.empty(N, K, dtype=torch.float32, device=g.device)
    gb = torch.empty(N, dtype=torch.float32, device=g.device) if want_gb else None
    _lib.check(lib.egtr_linear_backward_acc_f32(
        _stream(), g.data_ptr(), relu_out.data_ptr() if relu_out is not None else None, x2.data_ptr(), w.data_ptr(),
        float(alpha), gx.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else None, M, K, N,
        add1.data_ptr() if add1 is not None else None, add2.data_ptr() if add2 is not None else None),
        "egtr_linear_backward_acc_f32")
    return gx, gw, gb


class DecoderLayerTrainFunction(Function):
    """One Deformable-DETR decoder layer in TRAINING as a single autograd node (reference:
    DeformableDetrDecoderLayer.forward, model/deformable_detr.py:1390-1489; self-attention with the retained scaled-q / k maps
    :1107-1262; cross-attention :1026-1104 on the value projection handed in by ``DecoderValueProjTrainFunction``):

        q = s (x + pos) Wq^T + s bq,  k = (x + pos) Wk^T + bk,  v = x Wv^T + bv;   a = out_proj(softmax(q k^T) v)
        y1 = LN1(x + drop(a));   [off | logits] of (y1 + pos);   c = output_proj(MSDA(value, loc, softmax(logits)))
        y2 = LN2(y1 + drop(c));  y3 = LN3(y2 + drop(fc2(relu(fc1(y2)))))            returns (y3, q, k)

    The per-operation composition ran the same kernels as ~14 autograd nodes per layer; what autograd added around them -- per
    layer and step, on [B N, 256] tensors of 0.8 MB -- were the gradient-accumulation adds where branches meet (x feeds q / k, v
    and the residual; y1 feeds the offset / logit projections and the residual; y2 feeds fc1 and the residual; q and k also feed
    the relation head through the retained maps), ``x + pos`` / ``y1 + pos`` and their backward, and contiguous copies of views:
    ~18 launches of ~4 us each per layer (profiles/r06_train_ops_by_shape_before.txt: 55 add + 24 add_ + 14 copy_ + 12 mul
    launches of that shape per step).  Here every meeting point is the epilogue of the product that arrives last
    (egtr_linear_backward_acc_f32: up to two addends on grad_x; egtr_self_attn_backward_acc_f32: the maps' gradients on grad_q /
    grad_k); three ATen adds per layer remain (x + pos, y1 + pos, the two pos gradients).  Arithmetic: exactly the kernels of the
    composition (exact-f32 MFMA), so values agree to rounding of the changed summation order at the meeting points."""

    @staticmethod
    def forward(ctx, x, pos, ref, value, shapes, lsi, p_drop, masks, eps3, scaling, wq, bq, wk, bk, wv, bv, wo, bo, ln1w, ln1b,
                wso, bso, waw, baw, wop, bop, ln2w, ln2b, w1, b1, w2, b2, ln3w, ln3b):
        lib = _lib.lib()
        B, N, D = x.shape
        M = B * N
        dev = x.device
        x2 = _rows256(x.detach())
        pos3 = pos.detach()          # [B, N, D], usually a stride-0 batch expansion of the query table: added as it is (no copy)
        heads = 8
        d = [t.detach() for t in (wq, bq, wk, bk, wv, bv, wo, bo, ln1w, ln1b, wso, bso, waw, baw, wop, bop, ln2w, ln2b, w1, b1,
                                  w2, b2, ln3w, ln3b)]
        (wq_, bq_, wk_, bk_, wv_, bv_, wo_, bo_, g1, be1, wso_, bso_, waw_, baw_, wop_, bop_, g2, be2, w1_, b1_, w2_, b2_, g3,
         be3) = [t if t.is_contiguous() else t.contiguous() for t in d]
        p = float(p_drop)
        scale = 1.0 / (1.0 - p) if p > 0.0 else 1.0
        if masks is not None:
            m1, m2, m3 = masks[0], masks[1], masks[2]
        elif p > 0.0:
            mm = torch.empty(3, M, D, dtype=torch.uint8, device=dev).bernoulli_(1.0 - p)
            m1, m2, m3 = mm[0], mm[1], mm[2]
        else:
            m1 = m2 = m3 = None
        e1, e2, e3 = (float(v) for v in eps3)
        # ---- self-attention
        xp = _rows256(x2.view(B, N, D) + pos3)
        q = _skinny_fwd(xp, wq_, bq_, alpha=scaling)
        k = _skinny_fwd(xp, wk_, bk_)
        v = _skinny_fwd(x2, wv_, bv_)
        sa = torch.empty(M, D, dtype=torch.float32, device=dev)
        lse = torch.empty(B, heads, N, dtype=torch.float32, device=dev)
        _lib.check(lib.egtr_self_attn_forward_f32(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), B, N, heads, D // heads,
                                                  sa.data_ptr(), None, None, lse.data_ptr()), "egtr_self_attn_forward_f32")
        a = _skinny_fwd(sa, wo_, bo_)
        y1 = dropout_add_layernorm(a, x2, m1, scale, g1, be1, e1)
        # ---- cross-attention (MSDA over the encoder's value projection)
        y1p = _rows256(y1.view(B, N, D) + pos3)
        off = _skinny_fwd(y1p, wso_, bso_)
        lg = _skinny_fwd(y1p, waw_, baw_)
        L = shapes.shape[0]
        P_ = wso_.shape[0] // (heads * L * 2)
        refc = _chk(ref.detach().contiguous(), "reference_points", torch.float32)
        shp = _chk(shapes.contiguous(), "spatial_shapes", torch.int64)
        loc = torch.empty(B, N, heads, L, P_, 2, dtype=torch.float32, device=dev)
        attn = torch.empty(B, N, heads, L, P_, dtype=torch.float32, device=dev)
        _lib.check(lib.egtr_msda_geometry_forward_f32(_stream(), off.data_ptr(), off.stride(0), lg.data_ptr(), lg.stride(0),
                                                      refc.data_ptr(), refc.shape[-1], shp.data_ptr(), loc.data_ptr(),
                                                      attn.data_ptr(), M, heads, L, P_), "egtr_msda_geometry_forward_f32")
        val = value.detach()
        S = val.shape[1]
        val4 = (val if val.is_contiguous() else val.contiguous()).view(B, S, heads, D // heads)
        ca = _msda().ms_deform_attn_forward(val4, shp, lsi, loc, attn, 64).view(M, D)
        c = _skinny_fwd(ca, wop_, bop_)
        y2 = dropout_add_layernorm(c, y1, m2, scale, g2, be2, e2)
        # ---- feed-forward block
        h = _skinny_fwd(y2, w1_, b1_, relu=True)
        f = _skinny_fwd(h, w2_, b2_)
        y3 = dropout_add_layernorm(f, y2, m3, scale, g3, be3, e3)
        ctx.save_for_backward(x2, xp, q, k, v, sa, lse, a, m1, y1, y1p, off, refc, shp, lsi, loc, attn, val4, ca, c, m2, y2, h,
                              f, m3, wq_, wk_, wv_, wo_, g1, wso_, waw_, wop_, g2, w1_, w2_, g3)
        ctx.dims = (B, N, D, heads, L, P_, scale, e1, e2, e3, float(scaling))
        ctx.pos_shape = tuple(pos.shape)
        ctx.value_shape = tuple(value.shape)
        ctx.ref_needs_grad = bool(ref.requires_grad)
        return y3.view(B, N, D), q.view(B, N, D), k.view(B, N, D)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_y3, g_q_ext, g_k_ext):
        lib = _lib.lib()
        (x2, xp, q, k, v, sa, lse, a, m1, y1, y1p, off, refc, shp, lsi, loc, attn, val4, ca, c, m2, y2, h, f, m3, wq_, wk_, wv_,
         wo_, g1, wso_, waw_, wop_, g2, w1_, w2_, g3) = ctx.saved_tensors
        B, N, D, heads, L, P_, scale, e1, e2, e3, scaling = ctx.dims
        M = B * N
        dev = x2.device
        g3y = _rows256(g_y3)
        # ---- feed-forward block: LN3 <- fc2 <- ReLU <- fc1, the residual's gradient joins in fc1's data gradient
        gs3, gf, gbb3 = dropout_add_layernorm_backward(f, y2, m3, scale, g3, e3, g3y)
        g_h, d_w2, _ = _skinny_bwd(gf, h, w2_, want_gb=False)
        g_y2, d_w1, d_b1 = _skinny_bwd(g_h, y2, w1_, relu_out=h, add1=gs3)
        # ---- cross-attention
        gs2, gc, gbb2 = dropout_add_layernorm_backward(c, y1, m2, scale, g2, e2, g_y2)
        g_ca, d_wop, _ = _skinny_bwd(gc, ca, wop_, want_gb=False)
        g_value, g_loc, g_attn = _msda().ms_deform_attn_backward(val4, shp, lsi, loc, attn, g_ca.view(B, N, D), 64)
        g_off = torch.empty(M, off.shape[1], dtype=torch.float32, device=dev)
        g_lg = torch.empty(M, attn.numel() // M, dtype=torch.float32, device=dev)
        g_ref = torch.empty_like(refc) if ctx.ref_needs_grad else None
        _lib.check(lib.egtr_msda_geometry_backward_f32(
            _stream(), g_loc.data_ptr(), g_attn.data_ptr(), attn.data_ptr(), off.data_ptr(), off.stride(0), refc.data_ptr(),
            refc.shape[-1], shp.data_ptr(), g_off.data_ptr(), g_off.shape[1], g_lg.data_ptr(), g_lg.shape[1],
            g_ref.data_ptr() if g_ref is not None else None, M, heads, L, P_), "egtr_msda_geometry_backward_f32")
        t_so, d_wso, d_bso = _skinny_bwd(g_off, y1p, wso_)
        g_y1p, d_waw, d_baw = _skinny_bwd(g_lg, y1p, waw_, add1=t_so, out=t_so)          # d loss / d (y1 + pos)
        g_y1 = gs2.add_(g_y1p)
        # ---- self-attention
        gs1, ga, gbb1 = dropout_add_layernorm_backward(a, x2, m1, scale, g1, e1, g_y1)
        g_sa, d_wo, _ = _skinny_bwd(ga, sa, wo_, want_gb=False)
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        gqe = _rows256(g_q_ext) if g_q_ext is not None else None
        gke = _rows256(g_k_ext) if g_k_ext is not None else None
        _lib.check(lib.egtr_self_attn_backward_acc_f32(
            _stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), sa.data_ptr(), lse.data_ptr(), g_sa.data_ptr(), B, N, heads,
            D // heads, gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), gqe.data_ptr() if gqe is not None else None,
            gke.data_ptr() if gke is not None else None), "egtr_self_attn_backward_acc_f32")
        u_q, d_wq, d_bq = _skinny_bwd(gq, xp, wq_, alpha=scaling)
        g_xp, d_wk, d_bk = _skinny_bwd(gk, xp, wk_, add1=u_q, out=u_q)                  # d loss / d (x + pos)
        g_x, d_wv, d_bv = _skinny_bwd(gv, x2, wv_, add1=g_xp, add2=gs1, out=gs1)
        g_pos = (g_y1p + g_xp).view(B, N, D)
        if ctx.pos_shape != (B, N, D):
            g_pos = g_pos.sum_to_size(ctx.pos_shape)
        g_val = g_value.view(ctx.value_shape)
        return (g_x.view(B, N, D), g_pos, g_ref, g_val, None, None, None, None, None, None,
                d_wq, d_bq, d_wk, d_bk, d_wv, d_bv, d_wo, gbb1[512:768], gbb1[0:256], gbb1[256:512],
                d_wso, d_bso, d_waw, d_baw, d_wop, gbb2[512:768], gbb2[0:256], gbb2[256:512],
                d_w1, d_b1, d_w2, gbb3[512:768], gbb3[0:256], gbb3[256:512])


def decoder_layer_train_supported(layer, x, pos, ref, value, attention_mask, output_attentions):
    """The decoder-layer training node serves the reference's training configuration: fp32 on the GPU, object-query rows,
    d_model 256, 8 heads, 4 levels x 4 points, 2-d reference points (no box refinement), ReLU FFN, no attention maps / masks /
    attention dropout, a precomputed (already masked) value projection."""
    sa, ca = layer.self_attn, layer.encoder_attn
    eligible = (DECODER_TRAIN_FUSED and ENCODER_TRAIN_FUSED and SKINNY_BACKWARD_FUSED and torch.is_grad_enabled()
                and layer.training and not output_attentions and attention_mask is None and torch.is_tensor(x) and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 3 and x.shape[0] * x.shape[1] <= SKINNY_MAX_ROWS
                and torch.is_tensor(value))
    ok = (eligible and x.shape[-1] == 256 and pos is not None and pos.dtype == torch.float32 and pos.shape[-1] == 256
          and ref is not None and ref.dim() == 4 and ref.shape[-1] == 2 and ref.shape[2] == 4
          and value.dim() == 3 and value.shape[-1] == 256 and value.dtype == torch.float32
          and sa.num_heads == 8 and sa.embed_dim == 256 and sa.dropout == 0.0 and sa.q_proj.bias is not None
          and ca.n_heads == 8 and ca.n_levels == 4 and ca.n_points == 4 and ca.d_model == 256
          and layer.activation_fn is torch.nn.functional.relu and layer.activation_dropout == 0.0
          and layer.fc1.weight.shape[0] % 64 == 0 and tuple(layer.fc2.weight.shape) == (256, layer.fc1.weight.shape[0])
          and x.shape[1] <= 640 and 0.0 <= layer.dropout < 1.0)
    return _gate("decoder_layer_train", eligible, ok,
                 lambda: f"states {tuple(x.shape)}: the training node serves d_model 256, 8 heads, 4 x 4 sampling points, ReLU, "
                         "2-d reference points, <= 640 queries, no attention dropout")


def decoder_layer_train(layer, x, pos, ref, value, spatial_shapes, level_start_index, masks=None):
    """(y3, scaled q, k) of one decoder layer through DecoderLayerTrainFunction."""
    sa, ca = layer.self_attn, layer.encoder_attn
    return DecoderLayerTrainFunction.apply(
        x, pos, ref, value, spatial_shapes, level_start_index, layer.dropout, masks,
        (layer.self_attn_layer_norm.eps, layer.encoder_attn_layer_norm.eps, layer.final_layer_norm.eps), float(sa.scaling),
        sa.q_proj.weight, sa.q_proj.bias, sa.k_proj.weight, sa.k_proj.bias, sa.v_proj.weight, sa.v_proj.bias,
        sa.out_proj.weight, sa.out_proj.bias, layer.self_attn_layer_norm.weight, layer.self_attn_layer_norm.bias,
        ca.sampling_offsets.weight, ca.sampling_offsets.bias, ca.attention_weights.weight, ca.attention_weights.bias,
        ca.output_proj.weight, ca.output_proj.bias, layer.encoder_attn_layer_norm.weight, layer.encoder_attn_layer_norm.bias,
        layer.fc1.weight, layer.fc1.bias, layer.fc2.weight, layer.fc2.bias, layer.final_layer_norm.weight,
        layer.final_layer_norm.bias)



def cached_weights(owner, name, tensors, builder):
    """Derived constants of module weights (stacks, slices, concatenations), built once and rebuilt when a source tensor
    is replaced, moved or modified in place.  The cache lives ON the owning module (``owner._egtr_derived``), and an
    entry keeps strong references to its source tensors and compares them by identity, storage pointer and version
    counter -- a process-global table keyed by ``id(module)`` could hand one model's constants to a later model that
    happens to reuse the same ids and storage.  Writes through ``.data`` do not bump the version counter: call
    ``invalidate_derived(model)`` after such an edit."""
    cache = owner.__dict__.get("_egtr_derived")
    if cache is None:
        cache = {}
        object.__setattr__(owner, "_egtr_derived", cache)
    hit = cache.get(name)
    if hit is not None:
        srcs, key, val = hit
        if len(srcs) == len(tensors) and all(a is b for a, b in zip(srcs, tensors)) and \
                key == tuple((t.data_ptr(), t._version) for t in tensors):
            return val
    with torch.no_grad():
        val = builder()
    cache[name] = (list(tensors), tuple((t.data_ptr(), t._version) for t in tensors), val)
    return val


def invalidate_derived(model):
    """Drop every derived constant cached on ``model``'s modules (after an in-place edit through ``.data``)."""
    for m in model.modules():
        if "_egtr_derived" in m.__dict__:
            m.__dict__["_egtr_derived"].clear()
        if "_folded" in m.__dict__:
            m.__dict__["_folded"] = None
        m.__dict__.pop("_fold_full", None)


# ---- the environment switches of the package (round 6: seven route switches, down from twenty-four) ----------------------------
# Each selects between two SHIPPED routes that both have a use; everything else that used to be switchable from the environment
# is either gone (the decoder's tagged hand-over, the double-buffered split GEMM: measured slower) or a plain module attribute
# that only the switch-off twin tests patch (the route it turns off is the generic composition that CPU tensors and unsupported
# shapes take anyway).  tests/test_gpu_model.py::test_full_size_with_every_kept_switch_off_at_once_vs_reference runs the
# combination of all of them.
#   EGTR_DECODER_CLUSTER=0      decoder_fused.ENABLED       decoder layer: one launch per layer  ->  per-operation launches
#   EGTR_GEMM_SPLIT_BF16=0      ops.GEMM_SPLIT_BF16         token-sized linears: split-bf16 matrix cores  ->  vendor fp32 GEMM; also the
#                                                           fp32 backbone's own split-bf16 kernels (stem, 3x3 convolutions, bottleneck
#                                                           tails: backbone.STEM_FUSED / CONV2_X6 / CONV3_FUSED / CONV1_X6)  ->  MIOpen /
#                                                           vendor GEMM + passes
#   EGTR_REL_HEAD_SPLIT_BF16=0  ops.REL_HEAD_SPLIT_BF16     relation head at inference: split-bf16  ->  exact-f32 MFMA kernel
#   EGTR_FFN_FUSED=0            ops.FFN_FUSED               encoder FFN / layer tail row-panel kernels  ->  separate launches
#   EGTR_BACKBONE_NHWC=0        backbone.NHWC_F32 / _BF16   inference backbone channels-last  ->  NCHW
#   EGTR_ENCODER_TRAIN_FUSED=0  ops.ENCODER_TRAIN_FUSED     training nodes (encoder / decoder layer, values, LayerNorm)  ->  per-op autograd
#   EGTR_TOKEN_LINEAR=0         ops.TOKEN_LINEAR            token-sized linears under autograd: TokenLinearFunction  ->  F.linear
# Not route switches: EGTR_HIP_LIBRARY (another build of the library), EGTR_STRICT_FAST_PATH (below), EGTR_TRUST_CHECKPOINT_PICKLE.
ENV_ROUTE_SWITCHES = ("EGTR_DECODER_CLUSTER", "EGTR_GEMM_SPLIT_BF16", "EGTR_REL_HEAD_SPLIT_BF16", "EGTR_FFN_FUSED",
                      "EGTR_BACKBONE_NHWC", "EGTR_ENCODER_TRAIN_FUSED", "EGTR_TOKEN_LINEAR")

# ---- leaving a HIP fast path is never silent ------------------------------------------------------------------------------
# Every route from a hand-written kernel to an ATen composition (unsupported shape, misaligned operand, a device that
# refuses the cluster kernel) is counted here and announced ONCE per reason; EGTR_STRICT_FAST_PATH=1 (bench.py sets it)
# turns the first one into an error, so a benchmark number can not come from a deoptimised path.
FALLBACKS = {}
STRICT_FAST_PATH = os.environ.get("EGTR_STRICT_FAST_PATH", "0") == "1"


class FastPathError(RuntimeError):
    pass


def note_fallback(name, why=""):
    n = FALLBACKS.get(name, 0)
    FALLBACKS[name] = n + 1
    if STRICT_FAST_PATH:
        raise FastPathError(f"left the HIP fast path '{name}': {why}")
    if n == 0:
        import warnings
        warnings.warn(f"egtr_amd: leaving the HIP fast path '{name}' ({why}); counted in egtr_amd.ops.FALLBACKS",
                      RuntimeWarning, stacklevel=3)


def _gate(name, eligible, ok, why):
    """A fast-path predicate's verdict, announced when it turns a call away that the path exists for (``eligible``: right
    device, dtype, mode and size class -- an explicit EGTR_* switch or a CPU / bf16 / tiny call is not a fall-off)."""
    if eligible and not ok:
        note_fallback(name, why() if callable(why) else why)
    return bool(ok)


def inference_fast_path(x):
    """True when the launch-count optimisations (grouped linears, LayerNorm + position output, batched value
    projections) apply: fp32 tensors on the GPU and no autograd graph being recorded."""
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


# Decoder at inference: the residual-add + LayerNorm steps run as prologues of the skinny linears that consume them
# (DeferredLayerNorm below) instead of in launches of their own.  "0": stand-alone add_layernorm_256 launches.
DEFER_LAYERNORM = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


class DeferredLayerNorm:
    """y = LayerNorm(a + b) that has NOT been computed yet: the skinny linears that consume y apply it as a prologue
    (``linear_grouped`` items with ``x=<DeferredLayerNorm>``; egtr_linear_grouped_ln_f32) and the first such launch also
    stores y into ``.out``.  Replaces the decoder's stand-alone residual-add + LayerNorm launches (4.8 us each at 200 rows:
    launch floor) at inference.  ``materialize()`` runs the stand-alone kernel when no linear consumes y."""

    def __init__(self, a, b, ln, out=None):
        if a.shape != b.shape or a.shape[-1] != 256:
            raise ValueError("DeferredLayerNorm: two [.., 256] tensors")
        self.a, self.b, self.ln = a, b, ln
        self.out = out if out is not None else torch.empty_like(a)
        self.done = False        # .out holds y
        self.claimed = False     # a group of a launch being assembled will store y

    @property
    def shape(self):
        return self.a.shape

    @property
    def device(self):
        return self.a.device

    def materialize(self):
        if not self.done:
            add_layer_norm_into(self.a, self.b, self.ln, self.out)
            self.done = self.claimed = True
        return self.out


def add_layer_norm_into(x, residual, ln, out):
    """out = LayerNorm(x + residual) through the stand-alone kernel (egtr_add_layernorm_f32; inference, 256 channels)."""
    lib = _lib.lib()
    x2 = _chk(x.contiguous(), "x", torch.float32)
    r2 = _chk(residual.contiguous(), "residual", torch.float32)
    _chk(out, "out", torch.float32)
    if out.shape != x2.shape or x2.shape[-1] != 256:
        raise ValueError("add_layer_norm_into: out must have the shape of x, 256 channels")
    st = lib.egtr_add_layernorm_f32(_stream(), x2.data_ptr(), r2.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                    out.data_ptr(), x2.numel() // 256, 256, float(ln.eps))
    _lib.check(st, "egtr_add_layernorm_f32")
    return out


def linear_grouped(items):
    """Several independent skinny linears in ONE HIP launch (egtr_linear_grouped_ln_f32).  ``items`` is a list of dicts:
    x [.., K] (or a ``DeferredLayerNorm``: the LayerNorm runs as the layer's prologue, K = 256), w [N, K], b [N] or None,
    optional pos ([pos_rows, 256], added to a DeferredLayerNorm input after the LayerNorm), out (2-D view [rows, N] with
    unit inner stride: rows of a larger buffer), alpha_x (scale on x), alpha (scale after the bias), relu.  Returns the
    list of outputs ([.., N], or the given ``out`` views).  Inference only (no autograd)."""
    import ctypes
    lib = _lib.lib()
    G = len(items)
    if not 0 < G <= 16:
        raise ValueError("linear_grouped: 1..16 groups")
    x0 = items[0]["x"]
    K = x0.shape[-1]
    xs, ws, bs, ys, Ms, Ns, lds, ax, al, rl, outs, keep = [], [], [], [], [], [], [], [], [], [], [], []
    lres, lga, lbe, leps, lpos, lprows, lout = [], [], [], [], [], [], []
    any_ln = False
    for it in items:
        x, w, b = it["x"], it["w"], it.get("b")
        dln = x if isinstance(x, DeferredLayerNorm) else None
        if dln is not None and dln.done:
            x, dln = dln.out, None
        lead = tuple(x.shape[:-1])
        if dln is not None:
            any_ln = True
            x2 = _chk(dln.a.reshape(-1, K).contiguous(), "x", torch.float32)
            r2 = _chk(dln.b.reshape(-1, K).contiguous(), "residual", torch.float32)
            ga = _chk(dln.ln.weight.detach().contiguous(), "ln.weight", torch.float32)
            be = _chk(dln.ln.bias.detach().contiguous(), "ln.bias", torch.float32)
            pos = it.get("pos")
            p2 = _chk(pos.reshape(-1, K).contiguous(), "pos", torch.float32) if pos is not None else None
            first = not dln.claimed     # exactly one group of the launch stores the LayerNorm result
            dln.claimed = True
            o2 = _chk(dln.out.view(-1, K), "ln_out", torch.float32) if first else None
            keep += [r2, ga, be, p2, o2]
            lres.append(r2.data_ptr()); lga.append(ga.data_ptr()); lbe.append(be.data_ptr()); leps.append(float(dln.ln.eps))
            lpos.append(p2.data_ptr() if p2 is not None else None); lprows.append(p2.shape[0] if p2 is not None else 1)
            lout.append(o2.data_ptr() if o2 is not None else None)
        else:
            if it.get("pos") is not None:
                raise ValueError("linear_grouped: pos needs a DeferredLayerNorm input")
            x2 = _chk(x.reshape(-1, K).contiguous(), "x", torch.float32)
            lres.append(None); lga.append(None); lbe.append(None); leps.append(0.0); lpos.append(None); lprows.append(1)
            lout.append(None)
        w2 = _chk(w.detach().contiguous(), "w", torch.float32)
        b2 = _chk(b.detach().contiguous(), "b", torch.float32) if b is not None else None
        if w2.shape[1] != K or x.shape[-1] != K:
            raise ValueError("linear_grouped: all groups share K")
        M, N = x2.shape[0], w2.shape[0]
        out = it.get("out")
        if out is None:
            y2 = torch.empty(M, N, dtype=torch.float32, device=x2.device)
            outs.append(y2.view(*lead, N))
        else:
            if out.dim() != 2 or out.shape != (M, N) or out.stride(1) != 1:
                raise ValueError("linear_grouped: out must be a [rows, N] view with unit inner stride")
            y2 = out
            outs.append(out)
        keep += [x2, w2, b2, y2]
        xs.append(x2.data_ptr()); ws.append(w2.data_ptr()); bs.append(b2.data_ptr() if b2 is not None else None)
        ys.append(y2.data_ptr()); Ms.append(M); Ns.append(N); lds.append(y2.stride(0))
        ax.append(float(it.get("alpha_x", 1.0))); al.append(float(it.get("alpha", 1.0)))
        rl.append(1 if it.get("relu") else 0)
    PA, IA, FA = ctypes.c_void_p * G, ctypes.c_int * G, ctypes.c_float * G
    if any_ln:
        st = lib.egtr_linear_grouped_ln_f32(_stream(), G, PA(*xs), PA(*ws), PA(*bs), PA(*ys), IA(*Ms), IA(*Ns), IA(*lds),
                                            FA(*ax), FA(*al), IA(*rl), K, PA(*lres), PA(*lga), PA(*lbe), FA(*leps),
                                            PA(*lpos), IA(*lprows), PA(*lout))
        _lib.check(st, "egtr_linear_grouped_ln_f32")
        for it in items:
            if isinstance(it["x"], DeferredLayerNorm):
                it["x"].done = True
    else:
        st = lib.egtr_linear_grouped_f32(_stream(), G, PA(*xs), PA(*ws), PA(*bs), PA(*ys), IA(*Ms), IA(*Ns), IA(*lds),
                                         FA(*ax), FA(*al), IA(*rl), K)
        _lib.check(st, "egtr_linear_grouped_f32")
    return outs


def bias_relu_maxpool(x, bias):
    """relu(max_pool2d(x, 3, 2, 1) + bias[c]) in one HIP pass (== max_pool2d(relu(x + bias[c]), 3, 2, 1) bit for bit): the
    ResNet stem epilogue.  fp32 NCHW, inference only."""
    lib = _lib.lib()
    N, C, H, W_ = x.shape
    x2 = _chk(x.contiguous(), "x", torch.float32)
    b2 = _chk(bias.contiguous(), "bias", torch.float32)
    y = torch.empty(N, C, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, dtype=torch.float32, device=x.device)
    st = lib.egtr_bias_relu_maxpool3x3s2_f32(_stream(), x2.data_ptr(), b2.data_ptr(), y.data_ptr(), N, C, H, W_)
    _lib.check(st, "egtr_bias_relu_maxpool3x3s2_f32")
    return y


def box_decode(delta, init_reference, inter_references, eps=1e-5, logits_all=None):
    """sigmoid(delta + [inverse_sigmoid(reference_l), 0..]) for all decoder levels in one HIP launch (egr:286-305;
    reference_0 = init_reference, reference_l = inter_references[:, l-1]).  ``inter_references`` expanded from ONE tensor
    (stride 0 over the level axis: no box refinement) is not materialised.  With ``logits_all`` [B, Ld, N, C] the launch
    also returns argmax(logits_all[:, -1], -1) (the relation head's class lookup, egtr:405-413): (boxes, node_cls).
    Inference only."""
    lib = _lib.lib()
    B, Ld, N, four = delta.shape
    if four != 4:
        raise ValueError(f"delta must be [B, Ld, N, 4], got {tuple(delta.shape)}")
    d = _chk(delta.contiguous(), "delta", torch.float32)
    r0 = _chk(init_reference.contiguous(), "init_reference", torch.float32)
    RD = r0.shape[-1]
    # every level = the initial reference points, expanded over the level axis (the decoder without box refinement)
    same = (inter_references.dim() == 4 and Ld > 1 and inter_references.stride(1) == 0
            and inter_references.data_ptr() == init_reference.data_ptr()
            and inter_references.stride(0) == init_reference.stride(0)
            and tuple(inter_references.stride()[2:]) == tuple(init_reference.stride()[1:]))
    r1 = None if same else _chk(inter_references.contiguous(), "inter_references", torch.float32)
    if tuple(r0.shape) != (B, N, RD) or tuple(inter_references.shape) != (B, Ld, N, RD):
        raise ValueError(f"reference shapes {tuple(r0.shape)} / {tuple(inter_references.shape)} do not match delta "
                         f"{tuple(delta.shape)}")
    if RD not in (2, 4):
        raise ValueError(f"reference.shape[-1] should be 4 or 2, but got {RD}")
    out = torch.empty_like(d)
    lg, node, C = None, None, 0
    if logits_all is not None:
        lg = _chk(logits_all.contiguous(), "logits_all", torch.float32)
        C = lg.shape[-1]
        if tuple(lg.shape[:3]) != (B, Ld, N):
            raise ValueError("logits_all must be [B, Ld, N, C]")
        node = torch.empty(B, N, dtype=torch.int64, device=d.device)
    st = lib.egtr_box_decode_argmax_f32(_stream(), d.data_ptr(), r0.data_ptr(), r1.data_ptr() if r1 is not None else None,
                                        B, Ld, N, RD, float(eps), out.data_ptr(),
                                        lg.data_ptr() if lg is not None else None, C,
                                        node.data_ptr() if node is not None else None)
    _lib.check(st, "egtr_box_decode_argmax_f32")
    return out if logits_all is None else (out, node)


def bias_mask_rows_(y, bias, keep):
    """In place: y[g, r, :] = keep[r] ? y[g, r, :] + bias[g, :] : 0  (y [G, R, C]; keep [R] bool or None)."""
    lib = _lib.lib()
    G, R, C = y.shape
    _chk(y, "y", torch.float32)
    b2 = _chk(bias.detach().contiguous(), "bias", torch.float32)
    k2 = None
    if keep is not None:
        k2 = keep.reshape(-1).contiguous()
        k2 = k2.view(torch.uint8) if k2.dtype == torch.bool else k2.to(torch.uint8)
        _chk(k2, "keep")
    st = lib.egtr_bias_mask_rows_f32(_stream(), y.data_ptr(), b2.data_ptr(), k2.data_ptr() if k2 is not None else None,
                                     G, R, C)
    _lib.check(st, "egtr_bias_mask_rows_f32")
    return y


def add_layer_norm_pos(x, residual, ln, pos, out=None):
    """(ln(residual + x), ln(residual + x) + pos) in one HIP launch; pos is [rows_p, 256] with rows % rows_p == 0
    (broadcast over the batch).  ``out``: optional contiguous destination of the first result (e.g. a slice of the
    decoder's stacked intermediate states).  Inference only."""
    lib = _lib.lib()
    if x.dtype == torch.bfloat16:
        # bf16 model (stress configuration): same launch shape, bf16 storage, fp32 statistics
        x2 = _chk(x.contiguous(), "x", torch.bfloat16)
        r2 = _chk(residual.contiguous(), "residual", torch.bfloat16)
        p2 = _chk(pos.contiguous(), "pos", torch.bfloat16)
        _chk(ln.weight, "ln.weight", torch.bfloat16)
        rows, prow = x2.numel() // 256, p2.numel() // 256
        if x2.shape[-1] != 256 or rows % prow != 0 or out is not None:
            raise ValueError("add_layer_norm_pos (bf16): d_model must be 256, pos must tile the rows, no `out`")
        y, yp = torch.empty_like(x2), torch.empty_like(x2)
        st = lib.egtr_add_layernorm_pos_bf16(_stream(), x2.data_ptr(), r2.data_ptr(), ln.weight.data_ptr(),
                                             ln.bias.data_ptr(), y.data_ptr(), rows, 256, float(ln.eps), p2.data_ptr(),
                                             prow, yp.data_ptr())
        _lib.check(st, "egtr_add_layernorm_pos_bf16")
        return y, yp
    x2 = _chk(x.contiguous(), "x", torch.float32)
    r2 = _chk(residual.contiguous(), "residual", torch.float32)
    p2 = _chk(pos.contiguous(), "pos", torch.float32)
    rows = x2.numel() // 256
    prow = p2.numel() // 256
    if x2.shape[-1] != 256 or rows % prow != 0:
        raise ValueError("add_layer_norm_pos: d_model must be 256 and pos must tile the rows")
    y = torch.empty_like(x2) if out is None else _chk(out, "out", torch.float32)
    if y.shape != x2.shape:
        raise ValueError("add_layer_norm_pos: out must have the shape of x")
    yp = torch.empty_like(x2)
    st = lib.egtr_add_layernorm_pos_f32(_stream(), x2.data_ptr(), r2.data_ptr(), ln.weight.data_ptr(),
                                        ln.bias.data_ptr(), y.data_ptr(), rows, 256, float(ln.eps), p2.data_ptr(),
                                        prow, yp.data_ptr())
    _lib.check(st, "egtr_add_layernorm_pos_f32")
    return y, yp


# Token-sized fp32 linears (encoder: S ~ 12.5k rows) on the bf16 matrix cores through exact three-way operand splits
# (csrc/gemm_split.hip): fp32-level accuracy at 2.67x less matrix time than the fp32 MFMA / vendor fp32 GEMM.
# Inference (module_linear) and training (TokenLinearFunction: forward, data and weight gradients);
# EGTR_GEMM_SPLIT_BF16=0 keeps the vendor fp32 GEMM.
GEMM_SPLIT_BF16 = os.environ.get("EGTR_GEMM_SPLIT_BF16", "1") != "0"
# encoder at inference: the offsets / weights projection adds the position embeddings while loading its operand instead of
# reading a materialised `hidden + pos` written by the previous layer's epilogue ("0": materialise)
LAZY_POS = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6
GEMM_SPLIT_MIN_ROWS = 4096
GEMM_SPLIT_WGRAD = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


def gemm_split_weights(weight):
    """W [N, K] fp32 -> the operand stream of gemm_split_bf16_f32: [N/128][K/32][3 pieces][128][32] bf16."""
    N, K = weight.shape
    p = _split3_bf16(weight).view(3, N // 128, 128, K // 32, 32)
    return p.permute(1, 3, 0, 2, 4).contiguous()


def gemm_split_tile(weight, transposed=False):
    """``gemm_split_weights(weight)`` (``transposed``: of ``weight.t()``) in one launch (egtr_gemm_split_tile_weights_f32):
    the training step re-tiles each weight after every optimizer step, for the forward (W) and the data gradient (W^T)."""
    lib = _lib.lib()
    w = weight.detach()
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
        raise RuntimeError("gemm_split_tile: weight must be a 2-d float32 CUDA/HIP tensor with unit inner stride")
    N, K = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    out = torch.empty(N // 128, K // 32, 3, 128, 32, dtype=torch.bfloat16, device=w.device)
    st = lib.egtr_gemm_split_tile_weights_f32(_stream(), w.data_ptr(), w.stride(0), 1 if transposed else 0, N, K,
                                              out.data_ptr())
    _lib.check(st, "egtr_gemm_split_tile_weights_f32")
    return out


def gemm_split_tile_pair(weight):
    """(tiling of W, tiling of W^T) in one launch (egtr_gemm_split_tile_weights_pair_f32); N, K % 128 == 0."""
    lib = _lib.lib()
    w = weight.detach()
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
        raise RuntimeError("gemm_split_tile_pair: weight must be a 2-d float32 CUDA/HIP tensor with unit inner stride")
    N, K = w.shape
    out = torch.empty(2, 3 * N * K, dtype=torch.bfloat16, device=w.device)
    st = lib.egtr_gemm_split_tile_weights_pair_f32(_stream(), w.data_ptr(), w.stride(0), N, K, out.data_ptr())
    _lib.check(st, "egtr_gemm_split_tile_weights_pair_f32")
    return out[0].view(N // 128, K // 32, 3, 128, 32), out[1].view(K // 128, N // 32, 3, 128, 32)


def gemm_split_tile_pairs(weights):
    """[(tiling of W, tiling of W^T)] for up to 8 weights in ONE launch (egtr_gemm_split_tile_weights_multi_f32).  An entry
    is a [N, K] tensor or a pair (w_a, w_b) of tensors with the same K: the row-wise concatenation [w_a; w_b] tiled as one
    weight without materialising it.  N, K multiples of 128."""
    import ctypes
    lib = _lib.lib()
    n = len(weights)
    P, I = ctypes.c_void_p, ctypes.c_int
    w1, w2, ld1, ld2, split, Ns, Ks, outs = [], [], [], [], [], [], [], []
    for e in weights:
        a, b = (e if isinstance(e, (tuple, list)) else (e, None))
        a = a.detach()
        b = b.detach() if b is not None else None
        for t in (a, b):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1):
                raise RuntimeError("gemm_split_tile_pairs: 2-d float32 device tensors with unit inner stride expected")
        N, K = a.shape[0] + (b.shape[0] if b is not None else 0), a.shape[1]
        if b is not None and b.shape[1] != K:
            raise RuntimeError("gemm_split_tile_pairs: concatenated weights must share K")
        out = torch.empty(2, 3 * N * K, dtype=torch.bfloat16, device=a.device)
        w1.append(a.data_ptr()); ld1.append(a.stride(0)); split.append(a.shape[0])
        w2.append(b.data_ptr() if b is not None else None); ld2.append(b.stride(0) if b is not None else 0)
        Ns.append(N); Ks.append(K); outs.append(out)
    st = lib.egtr_gemm_split_tile_weights_multi_f32(
        _stream(), n, (P * n)(*w1), (I * n)(*ld1), (P * n)(*w2), (I * n)(*ld2), (I * n)(*split), (I * n)(*Ns), (I * n)(*Ks),
        (P * n)(*[o.data_ptr() for o in outs]))
    _lib.check(st, "egtr_gemm_split_tile_weights_multi_f32")
    return [(o[0].view(N // 128, K // 32, 3, 128, 32), o[1].view(K // 128, N // 32, 3, 128, 32))
            for o, N, K in zip(outs, Ns, Ks)]


def linear_split_bf16_wgrad(g, x):
    """g [M, N]^T . x [M, K] -> [N, K] (the weight gradient of a token-sized linear layer) through
    egtr_linear_split_bf16_wgrad_f32; unit inner strides, N, K % 128 == 0."""
    lib = _lib.lib()
    M, N = g.shape
    K = x.shape[1]
    if x.shape[0] != M or g.stride(1) != 1 or x.stride(1) != 1:
        raise RuntimeError("linear_split_bf16_wgrad: g [M, N] and x [M, K] with unit inner strides expected")
    ws = torch.empty(int(lib.egtr_linear_split_bf16_wgrad_workspace_floats(M, N, K)), dtype=torch.float32, device=g.device)
    gw = torch.empty(N, K, dtype=torch.float32, device=g.device)
    st = lib.egtr_linear_split_bf16_wgrad_f32(_stream(), g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0),
                                              gw.data_ptr(), ws.data_ptr(), M, N, K)
    _lib.check(st, "egtr_linear_split_bf16_wgrad_f32")
    return gw


def gemm_split_supported(x, N, K):
    rows = x.numel() // x.shape[-1]
    eligible = (GEMM_SPLIT_BF16 and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and rows >= GEMM_SPLIT_MIN_ROWS)
    return _gate("gemm_split", eligible, eligible and K % 32 == 0 and N % 128 == 0,
                 lambda: f"{rows} x {K} -> {N}: K must be a multiple of 32 and N of 128 (vendor GEMM instead)")


def linear_split_bf16(x, w_tiled, bias, N, relu=False, out=None):
    """act(x W^T + b) through egtr_linear_split_bf16_f32 (no autograd).  x [..., K] fp32 with unit inner stride and a
    uniform row stride (a column block of a wider buffer is fine); w_tiled from ``gemm_split_weights``; ``out``: optional
    contiguous [rows, N] fp32 destination."""
    lib = _lib.lib()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    _chk(w_tiled, "w_tiled", torch.bfloat16)
    if tuple(w_tiled.shape) != (N // 128, K // 32, 3, 128, 32):
        raise RuntimeError(f"w_tiled must be [{N // 128}, {K // 32}, 3, 128, 32], got {tuple(w_tiled.shape)}")
    b = _chk(bias.detach().contiguous(), "bias", torch.float32) if bias is not None else None
    if out is not None:
        y = _chk(out, "out", torch.float32)
        if tuple(y.shape) != (x2.shape[0], N):
            raise RuntimeError(f"out must be [{x2.shape[0]}, {N}], got {tuple(y.shape)}")
    else:
        y = torch.empty(x2.shape[0], N, dtype=torch.float32, device=x.device)
    st = lib.egtr_linear_split_bf16_f32(_stream(), x2.data_ptr(), x2.stride(0), w_tiled.data_ptr(),
                                        b.data_ptr() if b is not None else None, y.data_ptr(), N, x2.shape[0], K, N,
                                        1 if relu else 0)
    _lib.check(st, "egtr_linear_split_bf16_f32")
    return y if out is not None else y.view(*x.shape[:-1], N)


def linear_split_bf16_grouped(items):
    """Several token-sized linears with the same row count and K in ONE launch (egtr_linear_split_bf16_grouped_pos_f32).
    ``items``: dicts with x [..., K], wt (from ``gemm_split_weights``), N, optional b, relu, out ([rows, N] contiguous),
    pos ([pos_rows, K], added to x's rows (row % pos_rows) on the way into the kernel).  Returns the outputs ([rows, N]).
    Inference only."""
    import ctypes
    lib = _lib.lib()
    n = len(items)
    K = items[0]["x"].shape[-1]
    xs, outs, keep, poss = [], [], [], []
    for it in items:
        x2 = it["x"].reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        if not x2.is_cuda or x2.dtype != torch.float32:
            raise RuntimeError("linear_split_bf16_grouped: x must be a float32 CUDA/HIP tensor")
        N = int(it["N"])
        _chk(it["wt"], "wt", torch.bfloat16)
        if tuple(it["wt"].shape) != (N // 128, K // 32, 3, 128, 32):
            raise RuntimeError(f"wt must be [{N // 128}, {K // 32}, 3, 128, 32], got {tuple(it['wt'].shape)}")
        y = it.get("out")
        if y is None:
            y = torch.empty(x2.shape[0], N, dtype=torch.float32, device=x2.device)
        elif not (y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and tuple(y.shape) == (x2.shape[0], N)
                  and y.stride(1) == 1 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0):
            # (rows of a larger buffer are fine: the kernel takes the row stride)
            raise RuntimeError("linear_split_bf16_grouped: out must be a float32 [rows, N] view with unit inner stride, a "
                               "row stride that is a multiple of 4 and a 16-byte aligned base")
        b = it.get("b")
        if b is not None:
            b = _chk(b.detach().contiguous(), "bias", torch.float32)
        pos = it.get("pos")
        if pos is not None:
            pos = _chk(pos.reshape(-1, K).contiguous(), "pos", torch.float32)
            if x2.shape[0] % pos.shape[0]:
                raise RuntimeError("linear_split_bf16_grouped: pos must tile the rows")
        poss.append(pos)
        keep.append((x2, b))
        xs.append(x2)
        outs.append(y)
    M = xs[0].shape[0]
    if any(x2.shape[0] != M for x2 in xs):
        raise RuntimeError("linear_split_bf16_grouped: all inputs must have the same number of rows")
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    st = lib.egtr_linear_split_bf16_grouped_pos_f32(
        _stream(), n, PA(*[x2.data_ptr() for x2 in xs]), IA(*[x2.stride(0) for x2 in xs]),
        PA(*[it["wt"].data_ptr() for it in items]), PA(*[(b.data_ptr() if b is not None else None) for _, b in keep]),
        PA(*[y.data_ptr() for y in outs]), IA(*[y.stride(0) for y in outs]), IA(*[int(it["N"]) for it in items]),
        IA(*[1 if it.get("relu") else 0 for it in items]), M, K,
        PA(*[(p.data_ptr() if p is not None else None) for p in poss]), IA(*[(p.shape[0] if p is not None else 1) for p in poss]))
    _lib.check(st, "egtr_linear_split_bf16_grouped_pos_f32")
    return outs


# ---- the XS operand format (csrc/xs_format.h, csrc/xs_split.hip): weights of the row-panel kernels ----------------------
def xs_bytes(rows, K):
    return int(_lib.lib().egtr_xs_bytes(int(rows), int(K)))


def xs_split(x, pos=None, weights=False, plain=True):
    """fp32 [rows, K] (unit inner stride) -> XS(x): the exact three-way bf16 split in 1 KiB MFMA-operand fragments
    (csrc/xs_format.h; egtr_xs_split_f32), a flat uint8 tensor.  ``pos`` [pos_rows, K]: also XS(x + pos[row % pos_rows]);
    returns (XS(x) or None when ``plain`` is False, XS(x + pos)).  ``weights``: pieces rounded to nearest even."""
    lib = _lib.lib()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    if not x2.is_cuda or x2.dtype != torch.float32:
        raise RuntimeError("xs_split: x must be a float32 CUDA/HIP tensor")
    rows = x2.shape[0]
    n = xs_bytes(rows, K)
    if n == 0:
        raise RuntimeError(f"xs_split: K = {K} must be a multiple of 16")
    out = torch.empty(n, dtype=torch.uint8, device=x.device) if plain else None
    out_pos, p2 = None, None
    if pos is not None:
        p2 = _chk(pos.reshape(-1, K).contiguous(), "pos", torch.float32)
        out_pos = torch.empty(n, dtype=torch.uint8, device=x.device)
    st = lib.egtr_xs_split_f32(_stream(), x2.data_ptr(), x2.stride(0), p2.data_ptr() if p2 is not None else None,
                               p2.shape[0] if p2 is not None else 0, rows, K, out.data_ptr() if plain else None,
                               out_pos.data_ptr() if out_pos is not None else None, 1 if weights else 0)
    _lib.check(st, "egtr_xs_split_f32")
    return out if pos is None else (out, out_pos)


def conv1x1_tail_supported(a, N):
    """Shapes the bottleneck-tail kernel serves (csrc/conv_tail_x6.hip): fp32 pixel rows with unit inner stride, K = planes in
    {64, 128, 256, 512}, N a multiple of 64."""
    return (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1 and a.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0 and a.shape[1] in (64, 128, 256, 512) and N % 64 == 0)


def conv1x1_tail(a, a_shift, w_xs, bias, shortcut, N, relu_in=True, relu_out=True, tile=(0, 0)):
    """relu(relu(a + a_shift) W^T + bias + shortcut) in ONE HIP launch (egtr_conv1x1_tail_x6_f32): the last 1x1 convolution of
    a ResNet bottleneck on channels-last fp32 rows together with the shift + ReLU in front of it and the shift + shortcut +
    ReLU behind it (reference: model/deformable_detr.py:735-760, the timm ResNet-50 backbone with frozen batch norm).
    ``a`` [M, K] raw 3x3-convolution output, ``w_xs`` = ``xs_split(W [N, K], weights=True)``, ``shortcut`` [M, N] or None.
    fp32-level accuracy (six-term split-bf16 products).  Inference only."""
    lib = _lib.lib()
    M, K = a.shape
    for name, t in (("a_shift", a_shift), ("bias", bias), ("shortcut", shortcut)):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1):
            raise RuntimeError(f"conv1x1_tail: {name} must be a float32 device tensor with unit inner stride")
    y = torch.empty(M, N, dtype=torch.float32, device=a.device)
    st = lib.egtr_conv1x1_tail_x6_f32(
        _stream(), a.data_ptr(), a.stride(0), a_shift.data_ptr() if a_shift is not None else None, 1 if relu_in else 0,
        w_xs.data_ptr(), bias.data_ptr() if bias is not None else None,
        shortcut.data_ptr() if shortcut is not None else None, shortcut.stride(0) if shortcut is not None else 0,
        1 if relu_out else 0, y.data_ptr(), y.stride(0), M, K, N, int(tile[0]), int(tile[1]))
    _lib.check(st, "egtr_conv1x1_tail_x6_f32")
    return y


def stem_weights(w):
    """W [64, 3, 7, 7] fp32 (frozen BN scale folded in) -> the XS operand stream of the [64, 224] matrix the stem kernel walks:
    per kernel row 8 taps x 4 channels, the padded tap / channel zeros (csrc/stem_x6.hip)."""
    wm = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device=w.device)
    wm[:, :, :7, :3] = w.detach().permute(0, 2, 3, 1)
    return xs_split(wm.reshape(64, 224), weights=True)


def stem_weights_bf16(w):
    """W [64, 3, 7, 7] bf16 (frozen BN scale folded in) -> the packed MFMA-operand stream of the bf16 stem kernel
    (csrc/stem_bf16.hip): per kernel row 8 taps x 4 channels, the padded tap / channel zeros."""
    wm = torch.zeros(64, 7, 8, 4, dtype=torch.bfloat16, device=w.device)
    wm[:, :, :7, :3] = w.detach().permute(0, 2, 3, 1)
    return conv_tail_pack_bf16(wm.reshape(64, 224).contiguous())


def stem_fused_bf16_supported(x, w):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous()
            and tuple(w.shape) == (64, 3, 7, 7) and w.dtype == torch.bfloat16)


def stem_fused_bf16(x, w_packed, bias):
    """The bf16 twin of ``stem_fused`` (egtr_stem_conv7x7_pool_bf16): x [B, 3, H, W] NCHW bf16 -> channels-last bf16
    [B, 64, Hp, Wp]; ``bias`` fp32.  Inference only."""
    lib = _lib.lib()
    B, _, H, W = x.shape
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    y = torch.empty((B, 64, Hp, Wp), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    st = lib.egtr_stem_conv7x7_pool_bf16(_stream(), x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W)
    _lib.check(st, "egtr_stem_conv7x7_pool_bf16")
    return y


def stem_fused_supported(x, w):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous()
            and tuple(w.shape) == (64, 3, 7, 7))


def stem_fused(x, w_xs, bias):
    """maxpool3x3/2(relu(conv7x7/2(x) + bias)) of the ResNet stem in ONE HIP launch (egtr_stem_conv7x7_pool_x6_f32; reference:
    timm ResNet-50 conv1 -> bn1 -> act1 -> maxpool, model/deformable_detr.py:735-760).  x [B, 3, H, W] NCHW fp32 -> a
    channels-last [B, 64, Hp, Wp] tensor.  fp32-level accuracy (six-term split-bf16 products).  Inference only."""
    lib = _lib.lib()
    B, _, H, W = x.shape
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    y = torch.empty((B, 64, Hp, Wp), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    st = lib.egtr_stem_conv7x7_pool_x6_f32(_stream(), x.data_ptr(), w_xs.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W)
    _lib.check(st, "egtr_stem_conv7x7_pool_x6_f32")
    return y


def conv3x3_supported(x, N, stride=1, variant=0):
    """Shapes the split-bf16 3x3 convolution serves (csrc/conv3x3_x6.hip): channels-last fp32 [B, C, H, W] tensors (dense NHWC
    memory), padding 1, C == N in {64, 128, 256, 512} at stride 1 and {128, 256, 512} at stride 2."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and x.data_ptr() % 16 == 0
            and int(_lib.lib().egtr_conv3x3_phase_channels(int(x.shape[1]), int(N), int(stride), int(variant))) > 0)


def conv3x3_weights(w, stride=1, variant=0):
    """W [N, C, 3, 3] fp32 -> the XS operand stream of the [N, 9 C] matrix the kernel for (C, N, stride, variant) walks: channels
    in phases of CP (egtr_conv3x3_phase_channels), within a phase W[n][dy][dx][c'] (csrc/conv3x3_x6.hip)."""
    N, C = w.shape[:2]
    cp = int(_lib.lib().egtr_conv3x3_phase_channels(int(C), int(N), int(stride), int(variant)))
    if cp <= 0:
        raise RuntimeError(f"conv3x3_weights: C = {C}, N = {N}, stride {stride} is not served")
    wm = w.detach().reshape(N, C // cp, cp, 3, 3).permute(0, 1, 3, 4, 2).reshape(N, 9 * C).contiguous()
    return xs_split(wm, weights=True)


def conv3x3(x, w_xs, N, stride=1, variant=0):
    """3x3 convolution, stride 1 or 2, padding 1, no bias, on a channels-last fp32 tensor in ONE HIP launch with fp32-level
    accuracy on the bf16 matrix cores (egtr_conv3x3_x6_f32; reference: the timm ResNet-50 bottleneck's conv2,
    model/deformable_detr.py:735-760).  ``w_xs`` from ``conv3x3_weights`` with the same stride / variant.  Returns a channels-last
    [B, N, Ho, Wo] tensor.  Inference only."""
    lib = _lib.lib()
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, N, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    st = lib.egtr_conv3x3_x6_f32(_stream(), x.data_ptr(), w_xs.data_ptr(), y.data_ptr(), B, H, W, C, N, int(stride), int(variant))
    _lib.check(st, "egtr_conv3x3_x6_f32")
    return y


def conv1x1_strided_supported(x, N, stride):
    """Shapes the strided 1x1 convolution serves (csrc/conv3x3_x6.hip, one tap): channels-last fp32 [B, C, H, W] with C in
    {256, 512, 1024}, N a multiple of 128, stride 1 or 2."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and x.data_ptr() % 16 == 0 and x.shape[1] in (256, 512, 1024) and N % 128 == 0 and stride in (1, 2))


def conv1x1_strided(x, w_xs, N, stride):
    """1x1 convolution with stride (no padding, no bias) on a channels-last fp32 tensor in ONE HIP launch
    (egtr_conv1x1_strided_x6_f32): a bottleneck's shortcut projection; ``w_xs`` = ``xs_split(W [N, C], weights=True)``.  Returns
    the rows [B Ho Wo, N].  fp32-level accuracy.  Inference only."""
    lib = _lib.lib()
    B, C, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(B * Ho * Wo, N, dtype=torch.float32, device=x.device)
    st = lib.egtr_conv1x1_strided_x6_f32(_stream(), x.data_ptr(), w_xs.data_ptr(), y.data_ptr(), B, H, W, C, N, int(stride))
    _lib.check(st, "egtr_conv1x1_strided_x6_f32")
    return y


def conv1x1_tail_bf16_supported(a, N):
    """Shapes the bf16 bottleneck-tail kernel serves (csrc/conv_tail_bf16.hip): bf16 pixel rows with unit inner stride, K = planes
    in {64, 128, 256, 512}, N a multiple of 256."""
    return (a.is_cuda and a.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1 and a.stride(0) % 8 == 0
            and a.data_ptr() % 16 == 0 and a.shape[1] in (64, 128, 256, 512) and N % 256 == 0)


def conv_tail_pack_bf16(w):
    """W [N, K] bf16 -> the MFMA-operand stream of the bf16 bottleneck-tail kernel (egtr_conv1x1_tail_pack_weights_bf16)."""
    lib = _lib.lib()
    w = w.detach()
    if not w.is_cuda or w.dtype != torch.bfloat16 or w.dim() != 2 or w.stride(1) != 1:
        raise RuntimeError("conv_tail_pack_bf16: a 2-d bfloat16 device tensor with unit inner stride expected")
    N, K = w.shape
    out = torch.empty(N * K, dtype=torch.bfloat16, device=w.device)
    st = lib.egtr_conv1x1_tail_pack_weights_bf16(_stream(), w.data_ptr(), w.stride(0), N, K, out.data_ptr())
    _lib.check(st, "egtr_conv1x1_tail_pack_weights_bf16")
    return out


def conv1x1_tail_bf16(a, a_shift, w_packed, bias, shortcut, N, relu_in=True, relu_out=True):
    """relu(bf16(relu(a + a_shift) W^T) + bias + shortcut) in ONE HIP launch (egtr_conv1x1_tail_bf16): the bf16 twin of
    ``conv1x1_tail`` with the rounding points of the pass / GEMM / pass composition it replaces.  ``a`` [M, K] and ``shortcut``
    [M, N] bf16, shifts fp32, ``w_packed`` from ``conv_tail_pack_bf16``.  Inference only."""
    lib = _lib.lib()
    M, K = a.shape
    for name, t, dt in (("a_shift", a_shift, torch.float32), ("bias", bias, torch.float32), ("shortcut", shortcut, torch.bfloat16)):
        if t is not None and (not t.is_cuda or t.dtype != dt or t.stride(-1) != 1):
            raise RuntimeError(f"conv1x1_tail_bf16: {name} must be a {dt} device tensor with unit inner stride")
    y = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    st = lib.egtr_conv1x1_tail_bf16(
        _stream(), a.data_ptr(), a.stride(0), a_shift.data_ptr() if a_shift is not None else None, 1 if relu_in else 0,
        w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
        shortcut.data_ptr() if shortcut is not None else None, shortcut.stride(0) if shortcut is not None else 0,
        1 if relu_out else 0, y.data_ptr(), y.stride(0), M, K, N)
    _lib.check(st, "egtr_conv1x1_tail_bf16")
    return y


FFN_FUSED = os.environ.get("EGTR_FFN_FUSED", "1") != "0"


def ffn_fused_supported(x, fc1, fc2, ln):
    """The encoder layer's feed-forward block as one launch (csrc/ffn_x6.hip): fp32 inference, token-sized row counts,
    d_model = 256, hidden width a multiple of 64."""
    rows = x.numel() // x.shape[-1]
    eligible = FFN_FUSED and GEMM_SPLIT_BF16 and inference_fast_path(x) and rows >= GEMM_SPLIT_MIN_ROWS
    ok = (eligible and x.shape[-1] == 256 and tuple(fc1.weight.shape)[1] == 256 and fc1.weight.shape[0] % 64 == 0
          and tuple(fc2.weight.shape) == (256, fc1.weight.shape[0]) and fc1.bias is not None and fc2.bias is not None
          and ln.weight.shape[0] == 256 and fc1.weight.dtype == torch.float32)
    return _gate("ffn_fused", eligible, ok,
                 lambda: f"d_model {x.shape[-1]}, fc1 {tuple(fc1.weight.shape)}: the row-panel kernel serves d_model 256, "
                         "a hidden width that is a multiple of 64, biases present")


def ffn_fused(x, fc1, fc2, ln=None, pos=None):
    """LayerNorm(x + fc2(relu(fc1(x)))) [and that + pos] in ONE HIP launch (egtr_ffn_x6_f32; reference:
    model/deformable_detr.py:1335-1345 in eval mode); without ``ln``: fc2(relu(fc1(x))).  The [rows, ffn_dim] hidden
    activation never leaves the compute units.  Returns y or (y, y + pos).  Inference only."""
    lib = _lib.lib()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    rows, F = x2.shape[0], fc1.weight.shape[0]
    w1 = cached_weights(fc1, "xs_weight", [fc1.weight], lambda: xs_split(fc1.weight, weights=True))
    w2 = cached_weights(fc2, "xs_weight", [fc2.weight], lambda: xs_split(fc2.weight, weights=True))
    b1 = _chk(fc1.bias.detach().contiguous(), "fc1.bias", torch.float32)
    b2 = _chk(fc2.bias.detach().contiguous(), "fc2.bias", torch.float32)
    y = torch.empty(rows, K, dtype=torch.float32, device=x.device)
    g = bt = p2 = yp = None
    eps = 0.0
    if ln is not None:
        g = _chk(ln.weight.detach().contiguous(), "ln.weight", torch.float32)
        bt = _chk(ln.bias.detach().contiguous(), "ln.bias", torch.float32)
        eps = float(ln.eps)
        if pos is not None:
            p2 = _chk(pos.reshape(-1, K).contiguous(), "pos", torch.float32)
            if rows % p2.shape[0]:
                raise ValueError("ffn_fused: pos must tile the rows")
            yp = torch.empty_like(y)
    st = lib.egtr_ffn_x6_f32(_stream(), x2.data_ptr(), x2.stride(0), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                             b2.data_ptr(), g.data_ptr() if g is not None else None,
                             bt.data_ptr() if bt is not None else None, eps, p2.data_ptr() if p2 is not None else None,
                             p2.shape[0] if p2 is not None else 0, y.data_ptr(), yp.data_ptr() if yp is not None else None,
                             rows, K, F)
    _lib.check(st, "egtr_ffn_x6_f32")
    y = y.view(x.shape)
    return y if yp is None else (y, yp.view(x.shape))


# The whole tail of an encoder layer (output projection + LayerNorm + FFN block + LayerNorm) as ONE launch
# (egtr_encoder_tail_x6_f32); "0": projection + LayerNorm and the FFN block as two launches.
ENCODER_TAIL_FUSED = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


def encoder_tail_fused_supported(context, out_proj, ln1, fc1, fc2, ln2):
    return (ENCODER_TAIL_FUSED and proj_ln_fused_supported(context, out_proj, ln1)
            and ffn_fused_supported(context, fc1, fc2, ln2))


def encoder_tail_fused(context, hidden, out_proj, ln1, fc1, fc2, ln2, pos=None):
    """ln2(y1 + fc2(relu(fc1(y1)))) with y1 = ln1(hidden + out_proj(context)) [and the result + pos] in ONE HIP launch
    (egtr_encoder_tail_x6_f32; reference: model/deformable_detr.py:1102, 1326-1345 in eval mode).  ``context``: the
    deformable attention's output before its output projection.  Returns y or (y, y + pos).  Inference only."""
    lib = _lib.lib()
    K = context.shape[-1]

    def rows_of(t):
        t2 = t.reshape(-1, K)
        if t2.stride(1) != 1 or t2.stride(0) % 4 or t2.data_ptr() % 16:
            t2 = t2.contiguous()
        return t2

    c2, h2 = rows_of(context), rows_of(hidden)
    rows, F = c2.shape[0], fc1.weight.shape[0]
    if h2.shape[0] != rows:
        raise ValueError("encoder_tail_fused: context and hidden must have the same rows")
    wp = cached_weights(out_proj, "xs_weight", [out_proj.weight], lambda: xs_split(out_proj.weight, weights=True))
    w1 = cached_weights(fc1, "xs_weight", [fc1.weight], lambda: xs_split(fc1.weight, weights=True))
    w2 = cached_weights(fc2, "xs_weight", [fc2.weight], lambda: xs_split(fc2.weight, weights=True))
    f32 = [_chk(t.detach().contiguous(), n, torch.float32)
           for t, n in ((out_proj.bias, "out_proj.bias"), (ln1.weight, "ln1.weight"), (ln1.bias, "ln1.bias"),
                        (fc1.bias, "fc1.bias"), (fc2.bias, "fc2.bias"), (ln2.weight, "ln2.weight"), (ln2.bias, "ln2.bias"))]
    bp, g1, be1, b1, b2, g2, be2 = f32
    y = torch.empty(rows, K, dtype=torch.float32, device=context.device)
    p2 = yp = None
    if pos is not None:
        p2 = _chk(pos.reshape(-1, K).contiguous(), "pos", torch.float32)
        if rows % p2.shape[0]:
            raise ValueError("encoder_tail_fused: pos must tile the rows")
        yp = torch.empty_like(y)
    st = lib.egtr_encoder_tail_x6_f32(
        _stream(), c2.data_ptr(), c2.stride(0), h2.data_ptr(), h2.stride(0), wp.data_ptr(), bp.data_ptr(), g1.data_ptr(),
        be1.data_ptr(), float(ln1.eps), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g2.data_ptr(),
        be2.data_ptr(), float(ln2.eps), p2.data_ptr() if p2 is not None else None, p2.shape[0] if p2 is not None else 0,
        y.data_ptr(), yp.data_ptr() if yp is not None else None, rows, K, F)
    _lib.check(st, "egtr_encoder_tail_x6_f32")
    y = y.view(hidden.shape)
    return y if yp is None else (y, yp.view(hidden.shape))


def proj_ln_fused_supported(x, lin, ln):
    rows = x.numel() // x.shape[-1]
    eligible = FFN_FUSED and GEMM_SPLIT_BF16 and inference_fast_path(x) and rows >= GEMM_SPLIT_MIN_ROWS
    ok = (eligible and x.shape[-1] == 256 and tuple(lin.weight.shape) == (256, 256) and lin.bias is not None
          and ln.weight.shape[0] == 256 and lin.weight.dtype == torch.float32)
    return _gate("proj_ln_fused", eligible, ok, lambda: f"projection {tuple(lin.weight.shape)}: the kernel serves 256 -> 256 with a bias")


def proj_multi_fused_supported(x):
    rows = x.numel() // x.shape[-1]
    eligible = FFN_FUSED and GEMM_SPLIT_BF16 and inference_fast_path(x) and rows >= GEMM_SPLIT_MIN_ROWS
    return _gate("proj_multi_fused", eligible, eligible and x.shape[-1] == 256, lambda: f"d_model {x.shape[-1]} (256 served)")


def proj_ln_fused(x, lin, residual=None, ln=None, pos=None):
    """LayerNorm(residual + lin(x)) [and that + pos] for a 256 -> 256 nn.Linear in ONE HIP launch (egtr_proj_ln_x6_f32;
    reference: the attention output projection + residual + LayerNorm, model/deformable_detr.py:1102, 1326-1330); without
    ``ln``: lin(x).  Returns y or (y, y + pos).  Inference only."""
    lib = _lib.lib()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    w = cached_weights(lin, "xs_weight", [lin.weight], lambda: xs_split(lin.weight, weights=True))
    b = _chk(lin.bias.detach().contiguous(), "bias", torch.float32)
    y = torch.empty(rows, K, dtype=torch.float32, device=x.device)
    g = bt = p2 = yp = r2 = None
    eps = 0.0
    if ln is not None:
        r2 = residual.reshape(-1, K)
        if r2.stride(1) != 1 or r2.stride(0) % 4 or r2.data_ptr() % 16:
            r2 = r2.contiguous()
        g = _chk(ln.weight.detach().contiguous(), "ln.weight", torch.float32)
        bt = _chk(ln.bias.detach().contiguous(), "ln.bias", torch.float32)
        eps = float(ln.eps)
        if pos is not None:
            p2 = _chk(pos.reshape(-1, K).contiguous(), "pos", torch.float32)
            yp = torch.empty_like(y)
    st = lib.egtr_proj_ln_x6_f32(_stream(), x2.data_ptr(), x2.stride(0), w.data_ptr(), b.data_ptr(),
                                 r2.data_ptr() if r2 is not None else None, r2.stride(0) if r2 is not None else 0,
                                 g.data_ptr() if g is not None else None, bt.data_ptr() if bt is not None else None, eps,
                                 p2.data_ptr() if p2 is not None else None, p2.shape[0] if p2 is not None else 0,
                                 y.data_ptr(), yp.data_ptr() if yp is not None else None, rows, K)
    _lib.check(st, "egtr_proj_ln_x6_f32")
    y = y.view(x.shape)
    return y if yp is None else (y, yp.view(x.shape))


def proj_multi_fused(x, w_xs, num_weights, bias=None):
    """out[w] = x @ W_w^T (+ bias_w) for ``num_weights`` stacked 256 -> 256 weights applied to the same rows, ONE launch
    (egtr_proj_multi_x6_f32): ``w_xs`` = ``xs_split(torch.cat(weights, 0), weights=True)``.  Returns [num_weights, rows, 256].
    Inference only."""
    lib = _lib.lib()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    _chk(w_xs, "w_xs", torch.uint8)
    if w_xs.numel() != xs_bytes(num_weights * 256, 256):
        raise RuntimeError("proj_multi_fused: w_xs does not have the XS size of [num_weights * 256, 256]")
    b = _chk(bias.detach().contiguous(), "bias", torch.float32) if bias is not None else None
    out = torch.empty(num_weights, rows, K, dtype=torch.float32, device=x.device)
    st = lib.egtr_proj_multi_x6_f32(_stream(), x2.data_ptr(), x2.stride(0), w_xs.data_ptr(),
                                    b.data_ptr() if b is not None else None, out.data_ptr(), rows, K, num_weights)
    _lib.check(st, "egtr_proj_multi_x6_f32")
    return out


def module_linear(mod, x, alpha=1.0, relu=False):
    w = mod.weight
    if alpha == 1.0 and gemm_split_supported(x, w.shape[0], w.shape[1]):
        wt = cached_weights(mod, "gemm_split_bf16", [w], lambda: gemm_split_weights(w))
        return linear_split_bf16(x, wt, mod.bias, w.shape[0], relu)
    return linear(x, mod.weight, mod.bias, alpha, relu)


def scale_rows_multi(tensors, scales):
    """[t * s for t, s in zip(tensors, scales)] with s one value per row of t (shape [rows, 1, ...]), in one launch per 64
    tensors (egtr_scale_rows_multi_f32); tensors that do not qualify (not fp32 / columns not a multiple of 4) are multiplied by
    torch.  No autograd."""
    import ctypes
    lib = _lib.lib()
    out = [None] * len(tensors)
    todo = []
    for i, (t, s) in enumerate(zip(tensors, scales)):
        rows = t.shape[0]
        cols = t.numel() // rows if rows else 0
        if (t.is_cuda and t.dtype == torch.float32 and s.dtype == torch.float32 and s.numel() == rows and cols > 0
                and cols % 4 == 0):
            tc = t.contiguous()
            if tc.data_ptr() % 16 == 0:
                todo.append((i, tc, s.reshape(-1).contiguous(), rows, cols))
                continue
        out[i] = t * s
    for c0 in range(0, len(todo), 64):
        grp = todo[c0:c0 + 64]
        n = len(grp)
        res = [torch.empty_like(g[1]) for g in grp]
        PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
        st = lib.egtr_scale_rows_multi_f32(_stream(), n, PA(*[g[1].data_ptr() for g in grp]), PA(*[g[2].data_ptr() for g in grp]),
                                           PA(*[r.data_ptr() for r in res]), IA(*[g[3] for g in grp]), IA(*[g[4] for g in grp]))
        _lib.check(st, "egtr_scale_rows_multi_f32")
        for g, r in zip(grp, res):
            out[g[0]] = r
    return out


def bias_act_rows_(x2d, bias, residual=None, relu=True):
    """In-place y = act(x + bias[c] (+ residual)) on a channels-last activation given as its [rows, C] matrix (bf16 or fp32
    activations, fp32 bias: egtr_bias_act_nhwc_bf16 / _f32).  Inference only."""
    lib = _lib.lib()
    dt = x2d.dtype
    if dt not in (torch.bfloat16, torch.float32):
        raise TypeError("bias_act_rows_: bf16 or fp32 activations")
    _chk(x2d, "x", dt)
    _chk(bias, "bias", torch.float32)
    if residual is not None:
        _chk(residual, "residual", dt)
        if residual.shape != x2d.shape:
            raise ValueError("bias_act_rows_: residual must have the shape of x")
    rows, C = x2d.shape
    entry = "egtr_bias_act_nhwc_bf16" if dt == torch.bfloat16 else "egtr_bias_act_nhwc_f32"
    st = getattr(lib, entry)(_stream(), x2d.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                             x2d.data_ptr(), rows, C, 1 if relu else 0)
    _lib.check(st, entry)
    return x2d


def bias_act_(x, bias, residual=None, relu=True):
    """In-place y = act(x + bias[c] (+ residual)) on an NCHW activation (inference only, no autograd).  fp32, or bf16
    activations with an fp32 bias."""
    lib = _lib.lib()
    N, C, H, W_ = x.shape
    if x.dtype == torch.bfloat16:
        _chk(x, "x", torch.bfloat16)
        _chk(bias, "bias", torch.float32)
        if residual is not None:
            _chk(residual, "residual", torch.bfloat16)
        st = lib.egtr_bias_act_nchw_bf16(_stream(), x.data_ptr(), bias.data_ptr(),
                                         residual.data_ptr() if residual is not None else None, x.data_ptr(), N, C,
                                         H * W_, 1 if relu else 0)
        _lib.check(st, "egtr_bias_act_nchw_bf16")
        return x
    _chk(x, "x", torch.float32)
    _chk(bias, "bias", torch.float32)
    if residual is not None:
        _chk(residual, "residual", torch.float32)
    st = lib.egtr_bias_act_nchw_f32(_stream(), x.data_ptr(), bias.data_ptr(),
                                    residual.data_ptr() if residual is not None else None, x.data_ptr(), N, C, H * W_,
                                    1 if relu else 0)
    _lib.check(st, "egtr_bias_act_nchw_f32")
    return x


class BiasActFunction(Function):
    """y = relu(x + bias[c] (+ residual)) on an NCHW fp32 activation under autograd: one HIP pass forward
    (egtr_bias_act_nchw_f32), one mask pass backward (the masked gradient is the gradient of x AND of the residual)."""

    @staticmethod
    def forward(ctx, x, bias, residual):
        lib = _lib.lib()
        x = _chk(x.contiguous(), "x", torch.float32)
        _chk(bias, "bias", torch.float32)
        r = _chk(residual.contiguous(), "residual", torch.float32) if residual is not None else None
        N, C, H, W_ = x.shape
        y = torch.empty_like(x)
        st = lib.egtr_bias_act_nchw_f32(_stream(), x.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None,
                                        y.data_ptr(), N, C, H * W_, 1)
        _lib.check(st, "egtr_bias_act_nchw_f32")
        ctx.save_for_backward(y)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gm = torch.ops.aten.threshold_backward(g.contiguous(), y, 0.0)
        return gm, None, gm if ctx.has_residual else None


def bias_act(x, bias, residual=None):
    """relu(x + bias[c] (+ residual)) with autograd through x and residual (bias: a constant, e.g. a frozen-BN shift)."""
    return BiasActFunction.apply(x, bias, residual)


def sine_position_embedding(pixel_mask, embedding_dim, temperature, scale, eps=1e-6):
    """DeformableDetrSinePositionEmbedding(normalize=True) (dd:850-876) with the ~20 elementwise kernels after the two
    cumulative sums fused into one HIP kernel.  pixel_mask [B,H,W] bool/int -> [B, 2*embedding_dim, H, W] fp32."""
    lib = _lib.lib()
    y_embed = pixel_mask.cumsum(1, dtype=torch.float32).contiguous()
    x_embed = pixel_mask.cumsum(2, dtype=torch.float32).contiguous()
    dim_t = torch.arange(embedding_dim, dtype=torch.float32, device=pixel_mask.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / embedding_dim)
    B, H, W_ = pixel_mask.shape
    out = torch.empty(B, 2 * embedding_dim, H, W_, dtype=torch.float32, device=pixel_mask.device)
    st = lib.egtr_sine_pos_embed_f32(_stream(), y_embed.data_ptr(), x_embed.data_ptr(), dim_t.data_ptr(),
                                     out.data_ptr(), B, H, W_, embedding_dim, float(scale), float(eps))
    _lib.check(st, "egtr_sine_pos_embed_f32")
    return out


def input_proj_groupnorm_flatten(conv_outputs, input_projs):
    """Conv bias + GroupNorm + flatten(2).transpose(1, 2) + cat over the levels (dd:2209-2262) in two HIP launches.
    ``conv_outputs[l]``: bias-free output [B,256,H_l,W_l] of ``input_projs[l][0]`` (fp32, or bf16 for a bf16 model: bf16
    activations in and out, fp32 statistics); ``input_projs[l]`` = Sequential(Conv2d, GroupNorm).  Returns [B, S, 256].
    Inference only."""
    import ctypes
    lib = _lib.lib()
    L = len(conv_outputs)
    B, C = conv_outputs[0].shape[:2]
    gn0 = input_projs[0][1]
    dt = conv_outputs[0].dtype
    if dt not in (torch.float32, torch.bfloat16):
        raise TypeError("input_proj_groupnorm_flatten: fp32 or bf16 convolution outputs")
    xs = [_chk(x.contiguous(), "conv output", dt) for x in conv_outputs]
    for proj in input_projs[:L]:
        conv, gn = proj[0], proj[1]
        if gn.num_groups != gn0.num_groups or gn.eps != gn0.eps or conv.bias is None:
            raise ValueError("input_proj_groupnorm_flatten: levels must share the GroupNorm configuration")
    srcs = [t for proj in input_projs[:L] for t in (proj[0].bias, proj[1].weight, proj[1].bias)]
    if dt == torch.float32:
        keep = [tuple(_chk(t.detach().contiguous(), "input_proj parameter", torch.float32) for t in srcs[3 * l:3 * l + 3])
                for l in range(L)]
    else:   # the kernel takes fp32 parameters: widened once per parameter version
        flat = cached_weights(input_projs, "gn_params_f32", srcs,
                              lambda: [t.detach().float().contiguous() for t in srcs])
        keep = [tuple(flat[3 * l:3 * l + 3]) for l in range(L)]
    hw = [int(v) for x in xs for v in x.shape[-2:]]
    S = sum(h * w for h, w in zip(hw[0::2], hw[1::2]))
    out = torch.empty(B, S, C, dtype=dt, device=xs[0].device)
    stats = torch.empty(L * B * gn0.num_groups * 2, dtype=torch.float32, device=xs[0].device)
    PA, IA = ctypes.c_void_p * L, ctypes.c_int * (2 * L)
    entry = "egtr_input_proj_groupnorm_flatten_f32" if dt == torch.float32 else "egtr_input_proj_groupnorm_flatten_bf16"
    st = getattr(lib, entry)(
        _stream(), L, PA(*[x.data_ptr() for x in xs]), PA(*[k[0].data_ptr() for k in keep]),
        PA(*[k[1].data_ptr() for k in keep]), PA(*[k[2].data_ptr() for k in keep]), IA(*hw), B, C, gn0.num_groups,
        float(gn0.eps), stats.data_ptr(), out.data_ptr())
    _lib.check(st, entry)
    return out


def input_proj_groupnorm_tokens(token_outputs, input_projs):
    """The same for TOKEN-MAJOR projections [B, H_l*W_l, 256] (the channels-last backbone: the level's 1x1 convolution run as a
    plain GEMM, bias-free; bf16 or fp32): conv bias + GroupNorm(32) + concatenation in two launches, no transpose
    (egtr_input_proj_groupnorm_tokens_bf16 / _f32).  Returns [B, S, 256].  Inference only."""
    import ctypes
    lib = _lib.lib()
    L = len(token_outputs)
    B = token_outputs[0].shape[0]
    gn0 = input_projs[0][1]
    dt = token_outputs[0].dtype
    if dt not in (torch.bfloat16, torch.float32):
        raise TypeError("input_proj_groupnorm_tokens: bf16 or fp32 projections")
    xs = [_chk(x.contiguous(), "token-major projection", dt) for x in token_outputs]
    for proj, x in zip(input_projs[:L], xs):
        conv, gn = proj[0], proj[1]
        if gn.num_groups != 32 or gn.eps != gn0.eps or conv.bias is None or x.shape[-1] != 256 or x.shape[0] != B:
            raise ValueError("input_proj_groupnorm_tokens: 256 channels in 32 groups, one GroupNorm configuration")
    srcs = [t for proj in input_projs[:L] for t in (proj[0].bias, proj[1].weight, proj[1].bias)]
    flat = cached_weights(input_projs, "gn_params_f32", srcs, lambda: [t.detach().float().contiguous() for t in srcs])
    keep = [tuple(flat[3 * l:3 * l + 3]) for l in range(L)]
    toks = [int(x.shape[1]) for x in xs]
    S = sum(toks)
    out = torch.empty(B, S, 256, dtype=dt, device=xs[0].device)
    PA, IA = ctypes.c_void_p * L, ctypes.c_int * L
    stats = torch.empty(int(lib.egtr_input_proj_groupnorm_tokens_workspace_floats(L, IA(*toks), B)), dtype=torch.float32,
                        device=xs[0].device)
    entry = "egtr_input_proj_groupnorm_tokens_bf16" if dt == torch.bfloat16 else "egtr_input_proj_groupnorm_tokens_f32"
    st = getattr(lib, entry)(
        _stream(), L, PA(*[x.data_ptr() for x in xs]), PA(*[k[0].data_ptr() for k in keep]),
        PA(*[k[1].data_ptr() for k in keep]), PA(*[k[2].data_ptr() for k in keep]), IA(*toks), B, 256, 32, float(gn0.eps),
        stats.data_ptr(), out.data_ptr())
    _lib.check(st, entry)
    return out


_DIM_T = {}


def level_geometry(pixel_mask, spatial_shapes_list, level_embed, embedding_dim, temperature, scale, eps=1e-6):
    """Everything DeformableDetrModel.forward derives from ``pixel_mask`` alone, in one HIP kernel
    (egtr_level_geometry_f32): returns (mask_flatten [B,S] bool, lvl_pos_embed_flatten [B,S,2E] incl. level_embed,
    valid_ratios [B,L,2], encoder reference_points [B,S,L,2], mask bits [B, ceil(S/32)] int32 -- the mask packed one bit per
    token, which the model hands to the fused MSDA kernels as ``mask_bits``).  A bf16 ``level_embed`` (bf16 model) gives bf16 position rows
    rounded like the reference's composition (egtr_level_geometry_bf16); everything else stays fp32.  Inference only (no
    autograd through level_embed)."""
    import ctypes
    lib = _lib.lib()
    dev = pixel_mask.device
    key = (embedding_dim, float(temperature), str(dev))
    dim_t = _DIM_T.get(key)
    if dim_t is None:  # a constant of the module configuration (dd:864-865)
        dim_t = torch.arange(embedding_dim, dtype=torch.float32, device=dev)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / embedding_dim)
        _DIM_T[key] = dim_t
    if pixel_mask.dtype not in (torch.int64, torch.uint8, torch.bool):
        pixel_mask = (pixel_mask != 0).to(torch.uint8)
    pm = pixel_mask.contiguous()
    _chk(pm, "pixel_mask")
    pos_dtype = level_embed.dtype
    if pos_dtype == torch.bfloat16:
        le = level_embed.detach().float().contiguous()     # exact; [L, 2E]
    else:
        le = _chk(level_embed.detach().contiguous(), "level_embed", torch.float32)
    B, H, W_ = pm.shape
    L = len(spatial_shapes_list)
    S = sum(h * w for h, w in spatial_shapes_list)
    hw = (ctypes.c_int * (2 * L))(*[int(v) for hw_ in spatial_shapes_list for v in hw_])
    mask_u8 = torch.empty(B, S, dtype=torch.uint8, device=dev)
    bits = torch.empty(B, (S + 31) // 32, dtype=torch.int32, device=dev)
    pos = torch.empty(B, S, 2 * embedding_dim, dtype=pos_dtype, device=dev)
    vr = torch.empty(B, L, 2, dtype=torch.float32, device=dev)
    ref = torch.empty(B, S, L, 2, dtype=torch.float32, device=dev)
    entry = "egtr_level_geometry_bf16" if pos_dtype == torch.bfloat16 else "egtr_level_geometry_f32"
    st = getattr(lib, entry)(_stream(), pm.data_ptr(), pm.element_size(), dim_t.data_ptr(), le.data_ptr(), hw,
                             L, B, H, W_, embedding_dim, float(scale), float(eps), mask_u8.data_ptr(),
                             pos.data_ptr(), vr.data_ptr(), ref.data_ptr(), bits.data_ptr())
    _lib.check(st, entry)
    # `bits`: one bit per token, consumed by the fused MSDA kernels (kept in LDS there) -- returned, and handed down by the
    # model as an explicit `mask_bits` argument (until round 5 it travelled as a Python attribute on the mask tensor)
    return mask_u8.view(torch.bool), pos, vr, ref, bits


class LevelGeometryTrainFunction(Function):
    """``level_geometry`` under autograd: the position embeddings are sine(pixel_mask) + level_embed[level], so the only
    gradient is d level_embed[l] = sum over the tokens of level l (all images) of d pos -- four mask-weighted column sums over the
    [B * S, 256] gradient (egtr_weighted_column_sum_f32, ~13 us each) instead of autograd's cat-backward slices and generic
    `sum` reductions (161 us for the largest level).  Replaces, in training, the per-level mask interpolation, sine embedding,
    `+ level_embed`, cat, get_valid_ratio and get_reference_points compositions (model/deformable_detr.py:2195-2278, 1616-1648)
    by the two launches the inference path uses."""

    @staticmethod
    def forward(ctx, level_embed, pixel_mask, spatial_shapes_list, embedding_dim, temperature, scale):
        mask, pos, vr, ref, _ = level_geometry(pixel_mask, spatial_shapes_list, level_embed, embedding_dim, temperature, scale)
        ctx.shapes = (tuple(spatial_shapes_list), pos.shape[0])
        ctx.mark_non_differentiable(mask, vr, ref)
        return pos, mask, vr, ref

    @staticmethod
    @once_differentiable
    def backward(ctx, g_pos, g_mask, g_vr, g_ref):
        shapes, B = ctx.shapes
        g2 = _rows256(g_pos)
        return torch.stack([weighted_column_sum(g2, wl) for wl in _level_row_weights(shapes, B, g2.device)]), \
            None, None, None, None, None


# One-hot level membership of every token row ([B * S] floats per level): a constant of (level shapes, batch).  Multi-scale
# training meets a new shape combination in most batches, so the table is a small LRU (each entry is L x B*S floats on the
# device) and an entry is built ON the device from an arange -- no host tensor, no H2D copy, legal under stream capture.
_LEVEL_ROW_WEIGHTS = {}
_LEVEL_ROW_WEIGHTS_MAX = 4


def _level_row_weights(shapes, B, device):
    key = (shapes, B, str(device))
    w = _LEVEL_ROW_WEIGHTS.pop(key, None)
    if w is None:
        sizes = [h * w_ for h, w_ in shapes]
        S = sum(sizes)
        tok = torch.arange(B * S, device=device, dtype=torch.int32) % S      # token index within its image
        w, start = [], 0
        for n in sizes:   # (separate allocations: a row of one [L, B * S] tensor is 16-byte aligned only when B * S % 4 == 0)
            w.append(((tok >= start) & (tok < start + n)).to(torch.float32))
            start += n
        while len(_LEVEL_ROW_WEIGHTS) >= _LEVEL_ROW_WEIGHTS_MAX:
            _LEVEL_ROW_WEIGHTS.pop(next(iter(_LEVEL_ROW_WEIGHTS)))
    _LEVEL_ROW_WEIGHTS[key] = w   # (re-inserted last: most recently used)
    return w


def level_geometry_train(pixel_mask, spatial_shapes_list, level_embed, embedding_dim, temperature, scale):
    """(mask_flatten, lvl_pos_embed_flatten, valid_ratios, encoder reference points) with autograd through level_embed."""
    pos, mask, vr, ref = LevelGeometryTrainFunction.apply(level_embed, pixel_mask, list(spatial_shapes_list), embedding_dim,
                                                          temperature, scale)
    return mask, pos, vr, ref


class AddLayerNormFunction(Function):
    """LayerNorm(x + residual) over 256 channels in one pass (csrc/elementwise.hip).  Backward: one pass as well
    (egtr_add_layernorm_backward_f32: statistics recomputed from the saved inputs, gamma / beta gradients from
    per-workgroup partials in a fixed order)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        lib = _lib.lib()
        x2 = _chk(x.contiguous(), "x", torch.float32)
        r2 = _chk(residual.contiguous(), "residual", torch.float32)
        y = torch.empty_like(x2)
        st = lib.egtr_add_layernorm_f32(_stream(), x2.data_ptr(), r2.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                        y.data_ptr(), x2.numel() // x2.shape[-1], x2.shape[-1], float(eps))
        _lib.check(st, "egtr_add_layernorm_f32")
        ctx.eps = eps
        ctx.save_for_backward(x2, r2, weight, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        lib = _lib.lib()
        x, r, w, b = ctx.saved_tensors
        g = _chk(gy.contiguous(), "grad", torch.float32)
        rows = x.numel() // x.shape[-1]
        gs = torch.empty_like(x)
        ws = torch.empty(int(lib.egtr_add_layernorm_backward_workspace_floats(rows)), dtype=torch.float32, device=x.device)
        gwb = torch.empty(2, x.shape[-1], dtype=torch.float32, device=x.device)
        st = lib.egtr_add_layernorm_backward_f32(_stream(), x.data_ptr(), r.data_ptr(), w.data_ptr(), g.data_ptr(),
                                                 gs.data_ptr(), ws.data_ptr(), gwb.data_ptr(), rows, x.shape[-1],
                                                 float(ctx.eps))
        _lib.check(st, "egtr_add_layernorm_backward_f32")
        return gs, gs, gwb[0], gwb[1], None


def add_layer_norm(x, residual, ln):
    """ln(residual + x) for an nn.LayerNorm over d_model = 256."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256:
        return AddLayerNormFunction.apply(x, residual, ln.weight, ln.bias, ln.eps)
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == 256 and not torch.is_grad_enabled()
            and ln.weight.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16):
        # bf16 inference (stress configuration): one pass, fp32 statistics
        lib = _lib.lib()
        x2 = _chk(x.contiguous(), "x", torch.bfloat16)
        r2 = _chk(residual.contiguous(), "residual", torch.bfloat16)
        y = torch.empty_like(x2)
        st = lib.egtr_add_layernorm_bf16(_stream(), x2.data_ptr(), r2.data_ptr(), ln.weight.data_ptr(),
                                         ln.bias.data_ptr(), y.data_ptr(), x2.numel() // 256, 256, float(ln.eps))
        _lib.check(st, "egtr_add_layernorm_bf16")
        return y
    return ln(residual + x)


class RelationHeadFunction(Function):
    """Fused pairwise gate + gated sum + relation / connectivity MLPs (replaces model/egtr.py:366-416)."""

    @staticmethod
    def forward(ctx, gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist, node_cls,
                want_gate_mean):
        lib = _lib.lib()
        B, N, T = gate_q.shape
        Hd = w2r.shape[1]
        R = w3r.shape[0]
        tens = [gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c]
        names = ["gate_q", "gate_k", "uq", "uk", "b1", "w2r", "b2r", "w3r", "b3r", "w2c", "b2c", "w3c", "b3c"]
        tens = [_chk(t.contiguous(), n, torch.float32) for t, n in zip(tens, names)]
        rel = torch.empty(B, N, N, R, dtype=torch.float32, device=gate_q.device)
        conn = torch.empty(B, N, N, dtype=torch.float32, device=gate_q.device)
        gm = torch.zeros(T, dtype=torch.float32, device=gate_q.device) if want_gate_mean else None
        if triplet_dist is not None:
            _chk(triplet_dist, "triplet_dist", torch.float32)
            _chk(node_cls, "node_cls", torch.int64)
            c1 = triplet_dist.shape[0]
        else:
            c1 = 0
        need_grad = any(ctx.needs_input_grad[:13])
        P_ = B * N * N
        h1s = torch.empty(2, P_, Hd, dtype=torch.float32, device=gate_q.device) if need_grad else None
        h2s = torch.empty(2, P_, Hd, dtype=torch.float32, device=gate_q.device) if need_grad else None
        if REL_HEAD_TRAIN_X6 and need_grad and GEMM_SPLIT_BF16 and Hd == 256 and R <= 64 and T in (4, 7):
            # layers 2 and 3 on the bf16 matrix cores from split operands (the inference kernel with the two activation stores the
            # backward needs); the weight streams are rebuilt by one launch -- the weights change every step
            gq_, gk_, uq_, uk_, b1_, w2r_, b2r_, w3r_, b3r_, w2c_, b2c_, w3c_, b3c_ = tens
            w2xr, w3x, w2xc = rel_head_streams(w2r_, w3r_, w2c_)
            st = lib.egtr_rel_head_forward_bf16x6_save_f32(
                _stream(), gq_.data_ptr(), gk_.data_ptr(), uq_.data_ptr(), uk_.data_ptr(), b1_.data_ptr(), w2xr.data_ptr(),
                b2r_.data_ptr(), w3x.data_ptr(), b3r_.data_ptr(), w2xc.data_ptr(), b2c_.data_ptr(), w3c_.data_ptr(),
                b3c_.data_ptr(), triplet_dist.data_ptr() if triplet_dist is not None else None,
                node_cls.data_ptr() if triplet_dist is not None else None, B, N, T, Hd, R, c1, rel.data_ptr(),
                conn.data_ptr(), gm.data_ptr() if want_gate_mean else None, h1s.data_ptr(), h2s.data_ptr())
            _lib.check(st, "egtr_rel_head_forward_bf16x6_save_f32")
        else:
            st = lib.egtr_rel_head_forward_save_f32(
                _stream(), *[t.data_ptr() for t in tens], triplet_dist.data_ptr() if triplet_dist is not None else None,
                node_cls.data_ptr() if triplet_dist is not None else None, B, N, T, Hd, R, c1, rel.data_ptr(),
                conn.data_ptr(), gm.data_ptr() if want_gate_mean else None,
                h1s.data_ptr() if need_grad else None, h2s.data_ptr() if need_grad else None)
            _lib.check(st, "egtr_rel_head_forward_save_f32")
        if need_grad:
            ctx.save_for_backward(*tens, h1s, h2s)
        return rel, conn.unsqueeze(-1), gm

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_rel, grad_conn, grad_gm):
        """MLP part: rocBLAS GEMMs on the [B*N*N, 256] activations the forward kernel saved (no recomputation; the
        ReLU masks are applied with threshold_backward).  Pairwise part (gradients of the per-query tables and of
        the gate logits): HIP, egtr_rel_head_backward_pairs_f32.  The frequency bias is an additive constant."""
        (gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, h1s, h2s) = ctx.saved_tensors
        lib = _lib.lib()
        B, N, T = gate_q.shape
        Hd = w2r.shape[1]
        R = w3r.shape[0]
        P_ = B * N * N
        G = grad_rel.reshape(P_, R).contiguous()
        gc = grad_conn.reshape(P_, 1).contiguous()
        dh1 = torch.empty(2, P_, Hd, dtype=torch.float32, device=G.device)
        wgrad = (linear_split_bf16_wgrad if GEMM_SPLIT_BF16 and GEMM_SPLIT_WGRAD and Hd % 128 == 0
                 else (lambda g, h: g.t() @ h))
        # relation MLP (ReLU masks and bias gradients: one pass each, egtr_column_sum_f32, the mask applied in place)
        dh2, db2r = column_sum(G @ w3r, relu_output=h2s[0], inplace=True)
        dw3r = G.t() @ h2s[0]
        db3r = G.sum(0)   # 50 columns: the generic reduction is faster (19 vs 30 us)
        torch.mm(dh2, w2r, out=dh1[0])
        dw2r = wgrad(dh2, h1s[0])
        # connectivity MLP (one output)
        dh2, db2c = column_sum(gc * w3c, relu_output=h2s[1], inplace=True)
        dw3c = weighted_column_sum(h2s[1], gc).view(1, -1)   # gc^T h2: one pass instead of a 220 us GEMV
        db3c = gc.sum(0)
        torch.mm(dh2, w2c, out=dh1[1])
        dw2c = wgrad(dh2, h1s[1])
        del dh2
        db1 = torch.cat([column_sum(dh1[i], relu_output=h1s[i], inplace=True)[1] for i in range(2)])
        duq = torch.empty_like(uq)
        duk = torch.empty_like(uk)
        dgq = torch.empty_like(gate_q)
        dgk = torch.empty_like(gate_k)
        dz = torch.empty(P_ * T, dtype=torch.float32, device=G.device)
        st = lib.egtr_rel_head_backward_pairs_f32(_stream(), dh1.data_ptr(), gate_q.data_ptr(), gate_k.data_ptr(),
                                                  uq.data_ptr(), uk.data_ptr(), B, N, T, Hd, duq.data_ptr(),
                                                  duk.data_ptr(), dgq.data_ptr(), dgk.data_ptr(), dz.data_ptr())
        _lib.check(st, "egtr_rel_head_backward_pairs_f32")
        return (dgq, dgk, duq, duk, db1, dw2r, db2r, dw3r, db3r, dw2c, db2c, dw3c, db3c, None, None, None)


# bf16 model: layer 1 of the relation head (gated sum over the slots) on the matrix cores as well, tables pre-packed in
# operand order, W2 resident in LDS (csrc/rel_head_bf16.hip).  "0": the VALU layer 1 of rel_head_fwd_bf16w.
REL_HEAD_BF16_PACKED = True   # module attribute (tests patch it for the switch-off twin); no environment switch since round 6


def relation_head_bf16w(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist=None,
                        node_cls=None, want_gate_mean=False):
    """Inference forward of a bf16 model: the matrix products on the bf16 matrix cores with the bf16 parameters as they are
    (egtr_rel_head_forward_bf16p: all three layers, per-query tables uq / uk rounded to bf16 -- a bf16 model produces them
    in bf16 -- and packed in operand order by egtr_rel_head_pack_tables_bf16; or, with ops.REL_HEAD_BF16_PACKED = False,
    egtr_rel_head_forward_bf16w with an fp32 VALU layer 1); gates, biases and the outputs are fp32.  No autograd."""
    lib = _lib.lib()
    B, N, T = gate_q.shape
    Hd = w2r.shape[1]
    R = w3r.shape[0]
    dev = gate_q.device
    packed = REL_HEAD_BF16_PACKED and T <= 10 and Hd == 256 and R <= 64
    tables = []
    for t, n in ((uq, "uq"), (uk, "uk")):
        t = t.detach()
        if packed and t.dtype == torch.bfloat16:
            tables.append(_chk(t.contiguous(), n, torch.bfloat16))
        else:
            tables.append(_chk(t.float().contiguous(), n, torch.float32))
    f32 = [_chk(t.detach().float().contiguous(), n, torch.float32)
           for t, n in ((gate_q, "gate_q"), (gate_k, "gate_k"), (b1, "b1"), (b2r, "b2r"),
                        (b3r, "b3r"), (b2c, "b2c"), (b3c, "b3c"))]
    gq, gk, b1_, b2r_, b3r_, b2c_, b3c_ = f32
    wts = [_chk(t.detach().contiguous(), n, torch.bfloat16)
           for t, n in ((w2r, "w2r"), (w3r, "w3r"), (w2c, "w2c"), (w3c, "w3c"))]
    w2r_, w3r_, w2c_, w3c_ = wts
    rel = torch.empty(B, N, N, R, dtype=torch.float32, device=dev)
    conn = torch.empty(B, N, N, dtype=torch.float32, device=dev)
    gm = torch.zeros(T, dtype=torch.float32, device=dev) if want_gate_mean else None
    td = None
    c1 = 0
    if triplet_dist is not None:
        td = _chk(triplet_dist.detach().float().contiguous(), "triplet_dist", torch.float32)
        _chk(node_cls, "node_cls", torch.int64)
        c1 = td.shape[0]
    if packed:
        if tuple(tables[0].shape) != (B, N, T, 2 * Hd) or tuple(tables[1].shape) != (B, N, T, 2 * Hd):
            raise ValueError("relation_head_bf16w: uq / uk must be [B, N, T, 512]")
        pk = []
        for t in tables:   # [B * N] rows of [mlp 2][tile 8][half 2][channel 32][8 slots] bf16 = 16 KiB
            out = torch.empty(B * N, 8192, dtype=torch.bfloat16, device=dev)
            st = lib.egtr_rel_head_pack_tables_bf16(_stream(), t.data_ptr(), int(t.dtype == torch.bfloat16), B * N, T,
                                                    out.data_ptr())
            _lib.check(st, "egtr_rel_head_pack_tables_bf16")
            pk.append(out)
        st = lib.egtr_rel_head_forward_bf16p(
            _stream(), gq.data_ptr(), gk.data_ptr(), pk[0].data_ptr(), pk[1].data_ptr(), b1_.data_ptr(), w2r_.data_ptr(),
            b2r_.data_ptr(), w3r_.data_ptr(), b3r_.data_ptr(), w2c_.data_ptr(), b2c_.data_ptr(), w3c_.data_ptr(),
            b3c_.data_ptr(), td.data_ptr() if td is not None else None,
            node_cls.data_ptr() if td is not None else None, B, N, T, Hd, R, c1, rel.data_ptr(), conn.data_ptr(),
            gm.data_ptr() if want_gate_mean else None)
        _lib.check(st, "egtr_rel_head_forward_bf16p")
        return rel, conn.unsqueeze(-1), gm
    uq_, uk_ = tables
    st = lib.egtr_rel_head_forward_bf16w(
        _stream(), gq.data_ptr(), gk.data_ptr(), uq_.data_ptr(), uk_.data_ptr(), b1_.data_ptr(), w2r_.data_ptr(),
        b2r_.data_ptr(), w3r_.data_ptr(), b3r_.data_ptr(), w2c_.data_ptr(), b2c_.data_ptr(), w3c_.data_ptr(),
        b3c_.data_ptr(), td.data_ptr() if td is not None else None,
        node_cls.data_ptr() if td is not None else None, B, N, T, Hd, R, c1, rel.data_ptr(), conn.data_ptr(),
        gm.data_ptr() if want_gate_mean else None)
    _lib.check(st, "egtr_rel_head_forward_bf16w")
    return rel, conn.unsqueeze(-1), gm


# fp32 relation head on the bf16 matrix cores through three-way operand splits (csrc/rel_head.hip, rel_head_fwd_x6):
# fp32-level accuracy (tested against float64) at 2.67x less matrix time.  Inference only; set to False to run the
# exact-f32 MFMA kernel (v_mfma_f32_32x32x2_f32) everywhere.
REL_HEAD_SPLIT_BF16 = os.environ.get("EGTR_REL_HEAD_SPLIT_BF16", "1") != "0"


def _split3_bf16(w):
    """fp32 tensor -> [3, ...] bf16 pieces hi / mid / lo with hi + mid + lo == w to fp32 precision (round-to-nearest
    pieces; the residuals w - hi and (w - hi) - mid are exact in fp32)."""
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    r = w - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo])


def rel_head_split_weights(w2r, w3r, w2c):
    """The MFMA operand streams of rel_head_fwd_x6 (layouts documented in csrc/rel_head.hip):
    w2x [8 nt][16 t][3 piece][64 lane][8] per MLP, w3x [8 nt][2 kb][OT][3 piece][64 lane][8] (relation MLP)."""
    def w2x(w2):
        p = _split3_bf16(w2).view(3, 8, 32, 16, 2, 8)         # [piece, nt, pi, t, hf, e]
        return p.permute(1, 3, 0, 4, 2, 5).contiguous()        # [nt, t, piece, hf, pi, e]; lane = 32 hf + pi

    R = w3r.shape[0]
    OT = 1 if R <= 32 else 2
    w3p = torch.zeros(32 * OT, w3r.shape[1], dtype=torch.float32, device=w3r.device)
    w3p[:R] = w3r.detach().float()
    q = _split3_bf16(w3p).view(3, OT, 32, w3r.shape[1])       # [piece, ot, pi, n]
    dev = w3r.device
    nt = torch.arange(8, device=dev).view(8, 1, 1, 1)
    kb = torch.arange(2, device=dev).view(1, 2, 1, 1)
    hf = torch.arange(2, device=dev).view(1, 1, 2, 1)
    e = torch.arange(8, device=dev).view(1, 1, 1, 8)
    nidx = 32 * nt + 16 * kb + (e & 3) + 8 * (e >> 2) + 4 * hf  # [nt, kb, hf, e]
    g = q[:, :, :, nidx]                                        # [piece, ot, pi, nt, kb, hf, e]
    w3x = g.permute(3, 4, 1, 0, 5, 2, 6).contiguous()           # [nt, kb, ot, piece, hf, pi, e]
    return w2x(w2r), w3x, w2x(w2c)


def rel_head_streams(w2r, w3r, w2c):
    """``rel_head_split_weights`` as ONE launch (egtr_rel_head_streams_f32): the same three streams, bit for bit; what the
    training forward rebuilds after every optimizer step."""
    lib = _lib.lib()
    w2r_, w3r_, w2c_ = (_chk(t.detach().contiguous(), n, torch.float32)
                        for t, n in ((w2r, "w2r"), (w3r, "w3r"), (w2c, "w2c")))
    R = w3r_.shape[0]
    OT = 1 if R <= 32 else 2
    dev = w2r_.device
    w2xr = torch.empty(8, 16, 3, 2, 32, 8, dtype=torch.bfloat16, device=dev)
    w2xc = torch.empty(8, 16, 3, 2, 32, 8, dtype=torch.bfloat16, device=dev)
    w3x = torch.empty(8, 2, OT, 3, 2, 32, 8, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.egtr_rel_head_streams_f32(_stream(), w2r_.data_ptr(), w2c_.data_ptr(), w3r_.data_ptr(), w2r_.shape[1], R,
                                             w2xr.data_ptr(), w2xc.data_ptr(), w3x.data_ptr()),
               "egtr_rel_head_streams_f32")
    return w2xr, w3x, w2xc


def relation_head_split_bf16(gate_q, gate_k, uq, uk, b1, w2x_rel, b2r, w3x_rel, b3r, w2x_conn, b2c, w3c, b3c,
                             num_rel, triplet_dist=None, node_cls=None, want_gate_mean=False, sigmoid=False):
    """Inference forward, fp32 in / fp32 out, layers 2 and 3 on the bf16 matrix cores from split operands
    (egtr_rel_head_forward_bf16x6_f32; ``w2x_*`` / ``w3x_rel`` from ``rel_head_split_weights``).  No autograd."""
    lib = _lib.lib()
    B, N, T = gate_q.shape
    dev = gate_q.device
    f32 = [_chk(t.detach().contiguous(), n, torch.float32)
           for t, n in ((gate_q, "gate_q"), (gate_k, "gate_k"), (uq, "uq"), (uk, "uk"), (b1, "b1"), (b2r, "b2r"),
                        (b3r, "b3r"), (b2c, "b2c"), (w3c, "w3c"), (b3c, "b3c"))]
    gq, gk, uq_, uk_, b1_, b2r_, b3r_, b2c_, w3c_, b3c_ = f32
    R = int(num_rel)
    OT = 1 if R <= 32 else 2
    for t, n, shape in ((w2x_rel, "w2x_rel", (8, 16, 3, 2, 32, 8)), (w2x_conn, "w2x_conn", (8, 16, 3, 2, 32, 8)),
                        (w3x_rel, "w3x_rel", (8, 2, OT, 3, 2, 32, 8))):
        _chk(t, n, torch.bfloat16)
        if tuple(t.shape) != shape:
            raise RuntimeError(f"{n} must have shape {shape}, got {tuple(t.shape)}")
    rel = torch.empty(B, N, N, R, dtype=torch.float32, device=dev)
    conn = torch.empty(B, N, N, dtype=torch.float32, device=dev)
    gm = torch.zeros(T, dtype=torch.float32, device=dev) if want_gate_mean else None
    td = None
    c1 = 0
    if triplet_dist is not None:
        td = _chk(triplet_dist.detach().contiguous(), "triplet_dist", torch.float32)
        _chk(node_cls, "node_cls", torch.int64)
        c1 = td.shape[0]
    st = lib.egtr_rel_head_forward_bf16x6_f32(
        _stream(), gq.data_ptr(), gk.data_ptr(), uq_.data_ptr(), uk_.data_ptr(), b1_.data_ptr(), w2x_rel.data_ptr(),
        b2r_.data_ptr(), w3x_rel.data_ptr(), b3r_.data_ptr(), w2x_conn.data_ptr(), b2c_.data_ptr(), w3c_.data_ptr(),
        b3c_.data_ptr(), td.data_ptr() if td is not None else None,
        node_cls.data_ptr() if td is not None else None, B, N, T, 256, R, c1, rel.data_ptr(), conn.data_ptr(),
        gm.data_ptr() if want_gate_mean else None, 1 if sigmoid else 0)
    _lib.check(st, "egtr_rel_head_forward_bf16x6_f32")
    return rel, conn.unsqueeze(-1), gm


def relation_head(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist=None,
                  node_cls=None, want_gate_mean=False, owner=None, sigmoid=False):
    """``owner`` (optional nn.Module): where the derived split-bf16 weight streams of the inference kernel are cached.
    ``sigmoid``: return sigmoid(logits) (the model outputs) instead of the logits -- in the inference kernel's epilogue."""
    if sigmoid:
        rel, conn, gm = _relation_head(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist,
                                       node_cls, want_gate_mean, owner, True)
        return rel, conn, gm
    return _relation_head(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist, node_cls,
                          want_gate_mean, owner, False)


def _relation_head(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist, node_cls,
                   want_gate_mean, owner, sigmoid):
    if (REL_HEAD_SPLIT_BF16 and owner is not None and gate_q.dtype == torch.float32 and gate_q.is_cuda
            and w2r.shape == (256, 256) and w3r.shape[0] <= 64 and gate_q.shape[-1] <= 9
            and not (torch.is_grad_enabled() and any(
                t.requires_grad for t in (gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c)))):
        w2xr, w3xr, w2xc = cached_weights(owner, "rel_head_split_bf16", [w2r, w3r, w2c],
                                          lambda: rel_head_split_weights(w2r, w3r, w2c))
        return relation_head_split_bf16(gate_q, gate_k, uq, uk, b1, w2xr, b2r, w3xr, b3r, w2xc, b2c, w3c, b3c,
                                        w3r.shape[0], triplet_dist, node_cls, want_gate_mean, sigmoid)
    if sigmoid:   # every other kernel returns logits
        rel, conn, gm = _relation_head(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist,
                                       node_cls, want_gate_mean, None, False)
        return rel.sigmoid(), conn.sigmoid(), gm
    if gate_q.dtype == torch.bfloat16 and not (torch.is_grad_enabled() and any(
            t.requires_grad for t in (gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c))):
        rel, conn, gm = relation_head_bf16w(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c,
                                            triplet_dist, node_cls, want_gate_mean)
        return rel.to(torch.bfloat16), conn.to(torch.bfloat16), gm
    if gate_q.dtype != torch.float32:  # fp16 models / bf16 training: fp32 kernel, outputs cast back
        dt = gate_q.dtype
        f = [t.float() for t in (gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c)]
        rel, conn, gm = RelationHeadFunction.apply(*f, triplet_dist.float() if triplet_dist is not None else None,
                                                   node_cls, want_gate_mean)
        return rel.to(dt), conn.to(dt), gm
    return RelationHeadFunction.apply(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c,
                                      triplet_dist, node_cls, want_gate_mean)


@torch.no_grad()
def hungarian_match(logits, pred_boxes, targets, class_cost, bbox_cost, giou_cost, cost_min=None,
                    inverse_sigmoid_smoothing=None, cost_in=None, want_cost=False, want_status=False):
    """DeformableDetrHungarianMatcher.forward on the device (egtr_hungarian_match_f32): cost matrix + linear sum
    assignment per image in one launch, nothing copied to the host (the reference's ``.cpu()`` at dd:2985 is a device
    synchronisation per step).  ``targets``: list of dicts with "class_labels" / "boxes" (device tensors); their COUNTS
    are host integers (tensor shapes), so output shapes are static.  ``cost_min`` / ``inverse_sigmoid_smoothing``: the two
    fp32 scalars of the adaptive-smoothing offset (dd:2992-2998) or None.  ``cost_in``: per-image [N, T_b] cost matrices
    to solve instead (tests).  Returns (pred_idx, tgt_idx, match_cost) flat device tensors + the per-image counts
    [+ cost blocks] [+ status]: entries of image b are sorted by query index, i.e. scipy's output order."""
    lib = _lib.lib()
    if cost_in is not None:
        B = len(cost_in)
        N = cost_in[0].shape[0]
        dev = cost_in[0].device
        sizes = [int(c.shape[1]) for c in cost_in]
        cin = torch.cat([_chk(c.contiguous(), "cost_in", torch.float32).reshape(-1) for c in cost_in]) \
            if sum(sizes) else torch.zeros(1, device=dev)
        K = 0
        lg = bx = ti = tb = None
    else:
        B, N, K = logits.shape
        dev = logits.device
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        lg = _chk(logits.detach().contiguous(), "logits", torch.float32)
        bx = _chk(pred_boxes.detach().contiguous(), "pred_boxes", torch.float32)
        ti = torch.cat([t["class_labels"] for t in targets]).to(device=dev, dtype=torch.int64).contiguous()
        tb = torch.cat([t["boxes"] for t in targets]).to(device=dev, dtype=torch.float32).contiguous()
        cin = None
    n_out = [min(N, t) for t in sizes]
    offs = [0]
    for t in sizes:
        offs.append(offs[-1] + t)
    ooffs = [0]
    for t in n_out:
        ooffs.append(ooffs[-1] + t)
    meta = torch.tensor(offs + ooffs, dtype=torch.int32).to(dev, non_blocking=True)   # two small host -> device copies
    tot = max(ooffs[-1], 1)
    pred_idx = torch.empty(tot, dtype=torch.int64, device=dev)
    tgt_idx = torch.empty(tot, dtype=torch.int64, device=dev)
    mcost = torch.empty(tot, dtype=torch.float32, device=dev)
    cost_out = torch.empty(max(N * offs[-1], 1), dtype=torch.float32, device=dev) if want_cost else None
    status = torch.zeros(B, dtype=torch.int32, device=dev) if want_status else None
    smooth = 1 if cost_min is not None else 0
    n_scr = lib.egtr_hungarian_match_scratch_doubles(N, max(sizes) if sizes else 0, offs[-1])
    scratch = torch.empty(n_scr, dtype=torch.float64, device=dev) if n_scr > 0 else None
    st = lib.egtr_hungarian_match_f32(
        _stream(), lg.data_ptr() if lg is not None else None, bx.data_ptr() if bx is not None else None,
        ti.data_ptr() if ti is not None and ti.numel() else None, tb.data_ptr() if tb is not None and tb.numel() else None,
        meta.data_ptr(), meta.data_ptr() + 4 * (B + 1), B, N, K, max(sizes) if sizes else 0, float(class_cost),
        float(bbox_cost), float(giou_cost), smooth, float(cost_min) if smooth else 0.0,
        float(inverse_sigmoid_smoothing) if smooth else 0.0, pred_idx.data_ptr(), tgt_idx.data_ptr(), mcost.data_ptr(),
        cost_out.data_ptr() if want_cost else None, cin.data_ptr() if cin is not None else None,
        status.data_ptr() if want_status else None,
        scratch.data_ptr() if scratch is not None else None) if max(sizes, default=0) > 0 else 0
    _lib.check(st, "egtr_hungarian_match_f32")
    out = [pred_idx[:ooffs[-1]], tgt_idx[:ooffs[-1]], mcost[:ooffs[-1]], n_out]
    if want_cost:
        out.append([cost_out[N * offs[i]: N * offs[i + 1]].view(N, sizes[i]) for i in range(B)])
    if want_status:
        out.append(status)
    return tuple(out)


class RelationLossFunction(Function):
    """loss_rel / loss_connectivity of the SGG criterion (training mode, largest-score sampling) with their gradients
    from one pass over the logits (csrc/loss.hip, egtr_relation_loss_f32); backward only scales the stored gradients."""

    @staticmethod
    def forward(ctx, pred_rel, pred_conn, target_ptrs, pred_idx, tgt_idx, match_cost, out_off, nonmatching_cost,
                sample_negatives, sample_nonmatching):
        lib = _lib.lib()
        B, N, _, R = pred_rel.shape
        pr = _chk(pred_rel.detach().contiguous(), "pred_rel", torch.float32)
        pc = _chk(pred_conn.detach().contiguous(), "pred_connectivity", torch.float32)
        dev = pr.device
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        grad_rel = torch.empty_like(pr)
        grad_conn = torch.empty_like(pc)
        ws = torch.empty(int(lib.egtr_relation_loss_workspace_bytes(B, N)), dtype=torch.uint8, device=dev)
        st = lib.egtr_relation_loss_f32(_stream(), pr.data_ptr(), pc.data_ptr(), target_ptrs.data_ptr(),
                                        pred_idx.data_ptr(), tgt_idx.data_ptr(), match_cost.data_ptr(),
                                        out_off.data_ptr(), B, N, R, float(nonmatching_cost), int(sample_negatives),
                                        int(sample_nonmatching), loss.data_ptr(), grad_rel.data_ptr(),
                                        grad_conn.data_ptr(), ws.data_ptr())
        _lib.check(st, "egtr_relation_loss_f32")
        ctx.save_for_backward(grad_rel, grad_conn)
        return loss[0], loss[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_rel, g_conn):
        grad_rel, grad_conn = ctx.saved_tensors
        return grad_rel * g_rel, grad_conn * g_conn, None, None, None, None, None, None, None, None


class ClampNonFiniteFunction(Function):
    """dd:1346-1351 ("clamp the states iff any element is inf / nan") with the decision on the device: one reduction pass
    raises a flag, the in-place clamp and the gradient mask return at once while it is clear (csrc/elementwise.hip)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.lib()
        _chk(x, "hidden_states", torch.float32)
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
        cv = torch.finfo(torch.float32).max - 1000
        _lib.check(lib.egtr_any_nonfinite_f32(_stream(), x.data_ptr(), x.numel(), flag.data_ptr()),
                   "egtr_any_nonfinite_f32")
        _lib.check(lib.egtr_clamp_if_flag_f32(_stream(), x.data_ptr(), None, x.numel(), flag.data_ptr(), cv, 0),
                   "egtr_clamp_if_flag_f32")
        ctx.mark_dirty(x)
        ctx.save_for_backward(x, flag)
        ctx.cv = cv
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, flag = ctx.saved_tensors
        # Autograd forbids mutating a grad_output (the same buffer may be another branch's gradient: AddLayerNorm's backward
        # hands ONE tensor to x and to the residual), so the mask is applied to a private copy: one extra pass over the
        # states per encoder layer (~0.3 % of a step) for a result that is correct on the steps that did clamp.
        g = g.clone(memory_format=torch.contiguous_format)
        _lib.check(_lib.lib().egtr_clamp_if_flag_f32(_stream(), g.data_ptr(), x.data_ptr(), g.numel(), flag.data_ptr(),
                                                     ctx.cv, 1), "egtr_clamp_if_flag_f32")
        return g


def clamp_nonfinite_(x):
    """In place: x <- clamp(x, +-(finfo.max - 1000)) iff x holds an inf / nan (no host synchronisation).  fp32 contiguous
    device tensors; anything else takes the tensor composition of the same function."""
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.data_ptr() % 16 == 0:
        return ClampNonFiniteFunction.apply(x)
    bad = torch.logical_not(torch.isfinite(x).all())
    cv = torch.finfo(x.dtype).max - 1000
    return torch.where(bad, torch.clamp(x, min=-cv, max=cv), x)


class DetectionLossFunction(Function):
    """loss_ce / loss_bbox / loss_giou (+ the cardinality counts) of one output set with their gradients from one launch
    (csrc/loss.hip, egtr_detection_loss_f32); backward only scales the stored gradients."""

    @staticmethod
    def forward(ctx, logits, boxes, pred_idx, tgt_idx, match_off, tgt_labels, tgt_boxes, tgt_off, focal_alpha,
                num_boxes):
        lib = _lib.lib()
        B, N, C = logits.shape
        lg = _chk(logits.detach().contiguous(), "logits", torch.float32)
        bx = _chk(boxes.detach().contiguous(), "pred_boxes", torch.float32)
        dev = lg.device
        out = torch.empty(B, 4, dtype=torch.float32, device=dev)
        d_logits = torch.empty_like(lg)
        d_l1 = torch.empty_like(bx)
        d_giou = torch.empty_like(bx)
        st = lib.egtr_detection_loss_f32(_stream(), lg.data_ptr(), bx.data_ptr(), pred_idx.data_ptr(),
                                         tgt_idx.data_ptr(), match_off.data_ptr(), tgt_labels.data_ptr(),
                                         tgt_boxes.data_ptr(), tgt_off.data_ptr(), B, N, C, float(focal_alpha),
                                         float(num_boxes), d_logits.data_ptr(), d_l1.data_ptr(), d_giou.data_ptr(),
                                         out.data_ptr())
        _lib.check(st, "egtr_detection_loss_f32")
        ctx.save_for_backward(d_logits, d_l1, d_giou)
        sums = out.sum(0)
        card = out[:, 3]
        ctx.mark_non_differentiable(card)
        return sums[0], sums[1], sums[2], card

    @staticmethod
    @once_differentiable
    def backward(ctx, g_ce, g_bbox, g_giou, g_card):
        d_logits, d_l1, d_giou = ctx.saved_tensors
        return (d_logits * g_ce, d_l1 * g_bbox + d_giou * g_giou, None, None, None, None, None, None, None, None)


def pack_detection_targets(targets, device):
    """Concatenated class labels / boxes of a batch + per-image offsets, built once per step and shared by the output
    sets (main + auxiliary) of ``detection_losses``."""
    sizes = [int(t["class_labels"].shape[0]) for t in targets]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    if offs[-1]:
        labels = torch.cat([t["class_labels"] for t in targets]).to(device=device, dtype=torch.int64).contiguous()
        boxes = torch.cat([t["boxes"] for t in targets]).to(device=device, dtype=torch.float32).contiguous()
    else:   # keep the kernel's pointers valid
        labels = torch.zeros(1, dtype=torch.int64, device=device)
        boxes = torch.zeros(1, 4, dtype=torch.float32, device=device)
    toff = torch.tensor(offs, dtype=torch.int32).to(device, non_blocking=True)
    lengths = torch.tensor(sizes, dtype=torch.float32).to(device, non_blocking=True)
    return labels, boxes, toff, lengths


def detection_losses(logits, pred_boxes, flat_match, packed_targets, focal_alpha, num_boxes):
    """{"loss_ce", "loss_bbox", "loss_giou", "cardinality_error"} of one output set (egtr:611-712) from one launch.
    ``flat_match``: (pred_idx, tgt_idx, n_out) as the device matcher packs them; ``packed_targets``:
    ``pack_detection_targets``."""
    pred_idx, tgt_idx, n_out = flat_match
    labels, boxes, toff, lengths = packed_targets
    offs = [0]
    for n in n_out:
        offs.append(offs[-1] + int(n))
    moff = torch.tensor(offs, dtype=torch.int32).to(logits.device, non_blocking=True)
    if pred_idx.numel() == 0:
        pred_idx = torch.zeros(1, dtype=torch.int64, device=logits.device)
        tgt_idx = torch.zeros(1, dtype=torch.int64, device=logits.device)
    ce, bbox, giou, card = DetectionLossFunction.apply(logits, pred_boxes, pred_idx, tgt_idx, moff, labels, boxes, toff,
                                                       focal_alpha, num_boxes)
    return {"loss_ce": ce, "loss_bbox": bbox, "loss_giou": giou,
            "cardinality_error": (card - lengths).abs().mean()}


def relation_losses(pred_rel, pred_conn, targets, indices, matching_costs, nonmatching_cost, sample_negatives,
                    sample_nonmatching):
    """(loss_rel, loss_connectivity) for device tensors: see RelationLossFunction.  ``indices`` / ``matching_costs``: the
    matcher's per-image device tensors; ``targets[b]["rel"]``: dense fp32 [N, N, R] on the device."""
    dev = pred_rel.device
    rels = [_chk(t["rel"] if t["rel"].is_contiguous() else t["rel"].contiguous(), "target rel", torch.float32)
            for t in targets]
    ptrs = torch.tensor([r.data_ptr() for r in rels], dtype=torch.int64).to(dev, non_blocking=True)
    offs = [0]
    for src, _ in indices:
        offs.append(offs[-1] + int(src.shape[0]))
    out_off = torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True)
    pi = torch.cat([i[0] for i in indices]).to(device=dev, dtype=torch.int64)
    ti = torch.cat([i[1] for i in indices]).to(device=dev, dtype=torch.int64)
    mc = torch.cat(list(matching_costs)).to(device=dev, dtype=torch.float32)
    if pi.numel() == 0:   # keep the kernels' pointers valid
        pi = torch.zeros(1, dtype=torch.int64, device=dev)
        ti = torch.zeros(1, dtype=torch.int64, device=dev)
        mc = torch.zeros(1, dtype=torch.float32, device=dev)
    out = RelationLossFunction.apply(pred_rel, pred_conn, ptrs, pi, ti, mc, out_off, nonmatching_cost,
                                     sample_negatives, sample_nonmatching)
    del rels   # (kept alive until the launches were enqueued; the caller's targets own the storage)
    return out
