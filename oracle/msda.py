"""Oracle: multi-scale deformable attention (MSDA) sample + weighted sum, forward and backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Three independent restatements of the same op:

* ``msda_forward`` / ``msda_backward`` -- vectorised torch-CPU restatement of the arithmetic the reference's
  CUDA kernels perform (model/custom_kernel/cuda/ms_deform_im2col_cuda.cuh:33-84, 237-299 forward;
  87-159, 301-403 backward), same operation order: ``val = w1*v1 + w2*v2 + w3*v3 + w4*v4`` then
  ``out += val * attn`` accumulated over levels then points.
* ``msda_forward_grid_sample`` -- restatement of the reference's pure-PyTorch fallback
  ``ms_deform_attn_core_pytorch`` (model/deformable_detr.py:925-960): per-level ``F.grid_sample`` on
  ``2*loc-1`` (bilinear, zeros padding, align_corners=False) and a weighted sum.  This is what the reference
  executes when its CUDA extension is absent (deformable_detr.py:1096-1101), i.e. the CPU baseline.
* ``msda_forward_scalar`` -- scalar python loops, small cases only; one output element at a time.

Tensor conventions (reference: ms_deform_attn_cuda.cu:23-83):
  value [B, S, M, D]; spatial_shapes [L, 2] int64 (H, W); level_start_index [L] int64;
  sampling_loc [B, Lq, M, L, P, 2] (x, y) normalised to [0, 1]; attn [B, Lq, M, L, P]; out [B, Lq, M*D].
"""
import math

import torch
import torch.nn.functional as F


def _level_geometry(spatial_shapes, level_start_index=None):
    shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
    if level_start_index is None:
        starts, acc = [], 0
        for h, w in shapes:
            starts.append(acc)
            acc += h * w
    else:
        starts = [int(s) for s in level_start_index.tolist()]
    return shapes, starts


def _corner_terms(loc_l, H, W):
    """Per-sample bilinear geometry for one level (cuh:38-45, 268-288).

    loc_l [..., 2] -> dict with the sample-valid mask, integer corner coords, per-corner in-range masks and
    the four bilinear weights (w1..w4 in the kernel's naming: (y0,x0), (y0,x1), (y1,x0), (y1,x1))."""
    x = loc_l[..., 0] * W - 0.5
    y = loc_l[..., 1] * H - 0.5
    valid = (y > -1) & (x > -1) & (y < H) & (x < W)  # cuh:288
    y0 = torch.floor(y)
    x0 = torch.floor(x)
    lh = y - y0
    lw = x - x0
    hh = 1 - lh
    hw = 1 - lw
    y0 = y0.long()
    x0 = x0.long()
    y1 = y0 + 1
    x1 = x0 + 1
    m1 = valid & (y0 >= 0) & (x0 >= 0)  # cuh:55
    m2 = valid & (y0 >= 0) & (x1 <= W - 1)  # cuh:61
    m3 = valid & (y1 <= H - 1) & (x0 >= 0)  # cuh:67
    m4 = valid & (y1 <= H - 1) & (x1 <= W - 1)  # cuh:73
    return dict(valid=valid, y0=y0, x0=x0, y1=y1, x1=x1, lh=lh, lw=lw, hh=hh, hw=hw,
                masks=(m1, m2, m3, m4), weights=(hh * hw, hh * lw, lh * hw, lh * lw))


def _gather(value, base, yy, xx, W, mask):
    """value [B,S,M,D]; yy/xx/mask [B,Lq,M] -> [B,Lq,M,D], zero where mask is False (cuh:47-78)."""
    B, S, M, D = value.shape
    idx = base + yy.clamp(min=0) * W + xx.clamp(min=0)  # [B,Lq,M]
    idx = idx.clamp(0, S - 1)
    # gather along S for every (b, q, m): use value.permute -> [B,M,S,D]
    v = value.permute(0, 2, 1, 3)  # [B,M,S,D]
    idx_t = idx.permute(0, 2, 1)  # [B,M,Lq]
    g = torch.gather(v, 2, idx_t[..., None].expand(-1, -1, -1, D))  # [B,M,Lq,D]
    g = g.permute(0, 2, 1, 3)  # [B,Lq,M,D]
    return g * mask[..., None].to(value.dtype)


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn):
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    shapes, starts = _level_geometry(spatial_shapes, level_start_index)
    out = torch.zeros(B, Lq, M, D, dtype=value.dtype)
    for l, (H, W) in enumerate(shapes):  # cuh:274 loop over levels
        for p in range(P):  # cuh:283 loop over points
            t = _corner_terms(sampling_loc[:, :, :, l, p, :], H, W)
            ys = (t["y0"], t["y0"], t["y1"], t["y1"])
            xs = (t["x0"], t["x1"], t["x0"], t["x1"])
            val = None
            for k in range(4):  # cuh:82 w1*v1 + w2*v2 + w3*v3 + w4*v4
                vk = _gather(value, starts[l], ys[k], xs[k], W, t["masks"][k])
                term = t["weights"][k][..., None] * vk
                val = term if val is None else val + term
            out = out + val * attn[:, :, :, l, p][..., None]  # cuh:290
    return out.reshape(B, Lq, M * D)


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn, grad_output):
    """Analytic backward, following ms_deform_attn_col2im_bilinear (cuh:87-159) and the per-(b,q,m)
    reduction over the D channels of blocksize_aware_reduce_v1 (cuh:301-403).

    Returns (grad_value [B,S,M,D], grad_sampling_loc [B,Lq,M,L,P,2], grad_attn [B,Lq,M,L,P])."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    shapes, starts = _level_geometry(spatial_shapes, level_start_index)
    g = grad_output.reshape(B, Lq, M, D)
    grad_value = torch.zeros_like(value)
    grad_loc = torch.zeros_like(sampling_loc)
    grad_attn = torch.zeros_like(attn)
    gv_flat = grad_value.permute(0, 2, 1, 3).reshape(B * M * S, D)  # [B*M*S, D] (copy)
    gv_flat = torch.zeros_like(gv_flat)
    bm = (torch.arange(B)[:, None, None] * M + torch.arange(M)[None, None, :]).expand(B, Lq, M)
    for l, (H, W) in enumerate(shapes):
        for p in range(P):
            t = _corner_terms(sampling_loc[:, :, :, l, p, :], H, W)
            a = attn[:, :, :, l, p]
            top = g * a[..., None]  # top_grad * attn_weight (cuh:114)
            ys = (t["y0"], t["y0"], t["y1"], t["y1"])
            xs = (t["x0"], t["x1"], t["x0"], t["x1"])
            vs = [_gather(value, starts[l], ys[k], xs[k], W, t["masks"][k]) for k in range(4)]
            w = t["weights"]
            # grad_value: atomicAdd(grad_value[corner_k], w_k * top) for in-range corners (cuh:125-152)
            for k in range(4):
                idx = (starts[l] + ys[k].clamp(min=0) * W + xs[k].clamp(min=0)).clamp(0, S - 1)
                flat = (bm * S + idx).reshape(-1)
                contrib = (w[k][..., None] * top * t["masks"][k][..., None].to(value.dtype)).reshape(-1, D)
                gv_flat.index_add_(0, flat, contrib)
            hh, hw, lh, lw = t["hh"][..., None], t["hw"][..., None], t["lh"][..., None], t["lw"][..., None]
            v1, v2, v3, v4 = vs
            grad_h = -hw * v1 - lw * v2 + hw * v3 + lw * v4  # cuh:121-150 (masked corners contribute 0)
            grad_w = -hh * v1 + hh * v2 - lh * v3 + lh * v4
            val = w[0][..., None] * v1 + w[1][..., None] * v2 + w[2][..., None] * v3 + w[3][..., None] * v4
            grad_attn[:, :, :, l, p] = (g * val).sum(-1)  # cuh:156 + reduce 376-393
            grad_loc[:, :, :, l, p, 0] = (W * grad_w * top).sum(-1)  # cuh:157
            grad_loc[:, :, :, l, p, 1] = (H * grad_h * top).sum(-1)  # cuh:158
    grad_value = gv_flat.reshape(B, M, S, D).permute(0, 2, 1, 3).contiguous()
    return grad_value, grad_loc, grad_attn


def msda_forward_grid_sample(value, spatial_shapes, sampling_loc, attn):
    """Restatement of ms_deform_attn_core_pytorch (model/deformable_detr.py:925-960)."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    shapes, _ = _level_geometry(spatial_shapes)
    per_level = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_loc - 1  # :933
    sampled = []
    for l, (H, W) in enumerate(shapes):
        v_l = per_level[l].flatten(2).transpose(1, 2).reshape(B * M, D, H, W)  # :937-939
        g_l = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)  # [B*M, Lq, P, 2] :941
        sampled.append(F.grid_sample(v_l, g_l, mode="bilinear", padding_mode="zeros", align_corners=False))
    w = attn.transpose(1, 2).reshape(B * M, 1, Lq, L * P)  # :952-954
    out = (torch.stack(sampled, dim=-2).flatten(-2) * w).sum(-1).view(B, M * D, Lq)  # :955-959
    return out.transpose(1, 2).contiguous()


def msda_forward_scalar(value, spatial_shapes, level_start_index, sampling_loc, attn):
    """One output element at a time, python floats (double) -- small cases only (Appendix B of SURVEY.md)."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    shapes, starts = _level_geometry(spatial_shapes, level_start_index)
    v = value.double().numpy()
    loc = sampling_loc.double().numpy()
    a = attn.double().numpy()
    out = torch.zeros(B, Lq, M, D, dtype=torch.float64).numpy()
    for b in range(B):
        for q in range(Lq):
            for m in range(M):
                for l, (H, W) in enumerate(shapes):
                    for p in range(P):
                        x = loc[b, q, m, l, p, 0] * W - 0.5
                        y = loc[b, q, m, l, p, 1] * H - 0.5
                        if not (y > -1 and x > -1 and y < H and x < W):
                            continue
                        y0, x0 = math.floor(y), math.floor(x)
                        lh, lw = y - y0, x - x0
                        hh, hw = 1 - lh, 1 - lw
                        acc = 0.0
                        for (yy, xx, wgt) in ((y0, x0, hh * hw), (y0, x0 + 1, hh * lw),
                                              (y0 + 1, x0, lh * hw), (y0 + 1, x0 + 1, lh * lw)):
                            if 0 <= yy <= H - 1 and 0 <= xx <= W - 1:
                                acc = acc + wgt * v[b, starts[l] + yy * W + xx, m, :]
                        out[b, q, m, :] += acc * a[b, q, m, l, p]
    return torch.from_numpy(out).reshape(B, Lq, M * D)
