#!/usr/bin/env python3
"""Micro-benchmark of the MSDA kernels at the BASELINE shape (600x1000 -> S = 12537, M=8, D=32, L=4, P=4), with
sampling locations distributed like the model's: reference grid + the ring offset pattern of
DeformableDetrMultiscaleDeformableAttention._reset_parameters (+- 1..4 px per head) + jitter.

    python tools/msda_bench.py [--lq enc|dec] [--iters 200] [--bwd] [--jitter 0.5] [--batch 1]
Used under rocprofv3 (--kernel-trace --stats, and separate --pmc passes) for the roofline numbers in DESIGN.md.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(75, 125), (38, 63), (19, 32), (10, 16)]


def make_inputs(batch, lq_kind, jitter, dev, shapes=SHAPES, nq=200, dtype=torch.float32):
    g = torch.Generator().manual_seed(0)
    S = sum(h * w for h, w in shapes)
    M, D, L, P = 8, 32, len(shapes), 4
    value = torch.randn(batch, S, M, D, generator=g)
    if lq_kind == "enc":
        refs = []
        for (h, w) in shapes:
            ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w,
                                    indexing="ij")
            refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
        ref = torch.cat(refs, 0)  # [S,2]
    else:
        ref = torch.rand(nq, 2, generator=g)
    Lq = ref.shape[0]
    thetas = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    for i in range(P):
        grid[:, :, i, :] *= i + 1
    off = grid[None, None] + jitter * torch.randn(batch, Lq, M, L, P, 2, generator=g)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    attn = torch.softmax(torch.randn(batch, Lq, M, L * P, generator=g), -1).view(batch, Lq, M, L, P)
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    return [t.to(dev).contiguous() for t in (value.to(dtype), shp, lsi, loc, attn)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lq", default="enc")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--jitter", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="800x1333 shapes")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay 20 launches per HIP graph (no CPU launch floor)")
    ap.add_argument("--variant", type=int, default=0, help="kernel choice of the *_variant entries (include/egtr_hip_test.h)")
    ap.add_argument("--fused", action="store_true", help="fused-prologue entry (offsets | logits block + ref points)")
    ap.add_argument("--lib", default=None, help="load this build of libegtr_hip.so instead of the in-tree one (A/B of kernel builds)")
    a = ap.parse_args()
    if a.lib:
        from egtr_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    from egtr_amd.load_custom import load_hip_kernels
    k = load_hip_kernels()
    dev = "cuda:0"
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21)] if a.big else SHAPES
    value, shp, lsi, loc, attn = make_inputs(a.batch, a.lq, a.jitter, dev, shapes,
                                             dtype=torch.bfloat16 if a.bf16 else torch.float32)
    go = torch.randn(a.batch, loc.shape[1], 256, device=dev)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import hip_test_abi as TA     # the *_variant entries are test-only (include/egtr_hip_test.h)
    fn = ((lambda: TA.msda_backward_variant(value, shp, lsi, loc, attn, go, a.variant)) if a.variant
          else (lambda: k.ms_deform_attn_backward(value, shp, lsi, loc, attn, go, 64))) if a.bwd else \
        ((lambda: TA.msda_forward_variant(value, shp, lsi, loc, attn, a.variant))
         if (a.variant and not a.bf16) else (lambda: k.ms_deform_attn_forward(value, shp, lsi, loc, attn, 64)))
    if a.fused:
        B_, Lq_ = loc.shape[:2]
        shp_f = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev)
        refp = loc[:, :, 0, :, 0, :].clone()                               # [B, Lq, L, 2]
        offs = (loc - refp[:, :, None, :, None, :]) * shp_f[None, None, None, :, None, :]
        both = torch.cat([offs.reshape(B_, Lq_, 256), torch.log(attn.reshape(B_, Lq_, 128))], -1).contiguous()
        off_v = both[..., :256].view(B_, Lq_, 8, len(shapes), 4, 2)
        log_v = both[..., 256:].view(B_, Lq_, 8, 16)
        fn = lambda: k.ms_deform_attn_forward_fused(value, shp, lsi, off_v, log_v, refp, False, None)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    per = 1
    if a.graph:
        per = 20
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for _ in range(per):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        launch = gr.replay
        launch()
        torch.cuda.synchronize()
        fn_ = launch
    else:
        fn_ = fn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn_()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters / per
    B, S = value.shape[:2]
    Lq = loc.shape[1]
    e = value.element_size()
    # SURVEY.md 8(d), the ONE figure bench.py uses too: forward = value (capped by the gathered bytes) + loc + attn + out;
    # backward = the forward's READS + grad_out + 2 x grad_value (zero-init + accumulate) + grad_loc + grad_attn
    reads = B * (min(S * 256 * e, Lq * 8 * 16 * 4 * 32 * e) + Lq * 256 * 4 + Lq * 128 * 4)
    alg = reads + B * Lq * 256 * e
    if a.bwd:
        alg = reads + B * (Lq * 256 * 4 + 2 * S * 256 * 4 + Lq * 256 * 4 + Lq * 128 * 4)
    print(f"msda {'bwd' if a.bwd else 'fwd'} variant={a.variant} jitter={a.jitter} lq={a.lq} B={B} S={S} Lq={Lq} {'bf16' if a.bf16 else 'f32'}: "
          f"{us:.2f} us/launch (incl. the zero-fill of grad_value for bwd), algorithmic {alg / 1e6:.2f} MB -> "
          f"{alg / us / 1e3:.1f} GB/s = {alg / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
