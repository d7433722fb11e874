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
    ap.add_argument("--phases", action="store_true", help="print per-phase cycles of the tile kernel")
    ap.add_argument("--graph", action="store_true", help="replay 20 launches per HIP graph (no CPU launch floor)")
    ap.add_argument("--variant", type=int, default=0, help="forward kernel variant (include/egtr_hip.h)")
    ap.add_argument("--fused", action="store_true", help="fused-prologue entry (offsets | logits block + ref points)")
    a = ap.parse_args()
    from egtr_amd.load_custom import load_hip_kernels
    k = load_hip_kernels()
    dev = "cuda:0"
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21)] if a.big else SHAPES
    value, shp, lsi, loc, attn = make_inputs(a.batch, a.lq, a.jitter, dev, shapes,
                                             dtype=torch.bfloat16 if a.bf16 else torch.float32)
    go = torch.randn(a.batch, loc.shape[1], 256, device=dev)
    fn = (lambda: k.ms_deform_attn_backward(value, shp, lsi, loc, attn, go, 64, a.variant)) if a.bwd else \
        ((lambda: k.ms_deform_attn_forward_variant(value, shp, lsi, loc, attn, a.variant))
         if (a.variant and not a.bf16) else (lambda: k.ms_deform_attn_forward(value, shp, lsi, loc, attn, 64)))
    if a.fused:
        B_, Lq_ = loc.shape[:2]
        shp_f = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev)
        refp = loc[:, :, 0, :, 0, :].clone()                               # [B, Lq, L, 2]
        offs = (loc - refp[:, :, None, :, None, :]) * shp_f[None, None, None, :, None, :]
        both = torch.cat([offs.reshape(B_, Lq_, 256), torch.log(attn.reshape(B_, Lq_, 128))], -1).contiguous()
        off_v = both[..., :256].view(B_, Lq_, 8, len(shapes), 4, 2)
        log_v = both[..., 256:].view(B_, Lq_, 8, 16)
        fn = lambda: k.ms_deform_attn_forward_fused(value, shp, lsi, off_v, log_v, refp, False, None, variant=a.variant)
    if a.phases and a.variant in (11, 13):
        from egtr_amd import _lib
        cyc = torch.zeros(8, dtype=torch.int64, device=dev)
        out = torch.empty(a.batch, loc.shape[1], 256, device=dev)
        for _ in range(3):
            cyc.zero_()
            st = _lib.lib().egtr_msda_win_phase_cycles(torch.cuda.current_stream().cuda_stream, value.data_ptr(),
                                                       shp.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
                                                       attn.data_ptr(), a.batch, value.shape[1], shp.shape[0],
                                                       loc.shape[1], 4, 3 if a.variant == 11 else 5, out.data_ptr(), cyc.data_ptr())
            _lib.check(st, "phase cycles")
            torch.cuda.synchronize()
        c = cyc.tolist()
        n = max(c[7], 1)
        print(f"pipelined window kernel, wave 0, ticks per item: barrierX={c[0]/n:.0f} geometry={c[1]/n:.0f} "
              f"barrierY={c[2]/n:.0f} pack+copy+records={c[3]/n:.0f} gather={c[4]/n:.0f} dma_wait={c[5]/n:.0f}; "
              f"items={c[7]}; ticks per workgroup={c[6]/(512 if a.variant == 11 else 1024):.0f}")
    elif a.phases and a.variant in (8, 9, 10):
        from egtr_amd import _lib
        cyc = torch.zeros(6, dtype=torch.int64, device=dev)
        out = torch.empty(a.batch, loc.shape[1], 256, device=dev)
        for _ in range(3):
            cyc.zero_()
            st = _lib.lib().egtr_msda_win_phase_cycles(torch.cuda.current_stream().cuda_stream, value.data_ptr(),
                                                       shp.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
                                                       attn.data_ptr(), a.batch, value.shape[1], shp.shape[0],
                                                       loc.shape[1], 4, a.variant - 8, out.data_ptr(), cyc.data_ptr())
            _lib.check(st, "phase cycles")
            torch.cuda.synchronize()
        c = cyc.tolist()
        print(f"window kernel phases per work item (s_memtime ticks = 100 MHz?): P0={c[0]/c[4]:.0f} A={c[1]/c[4]:.0f} "
              f"B={c[2]/c[4]:.0f} C={c[3]/c[4]:.0f} items={c[4]} staged levels/item={c[5]/c[4]:.2f}")
    elif a.phases and a.variant in (5, 6):
        from egtr_amd import _lib
        cyc = torch.zeros(8, dtype=torch.int64, device=dev)
        out = torch.empty(a.batch, loc.shape[1], 256, device=dev)
        for _ in range(3):
            cyc.zero_()
            st = _lib.lib().egtr_msda_lane_phase_cycles(torch.cuda.current_stream().cuda_stream, value.data_ptr(),
                                                        shp.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
                                                        attn.data_ptr(), a.batch, value.shape[1], loc.shape[1],
                                                        a.variant - 5, out.data_ptr(), cyc.data_ptr())
            _lib.check(st, "phase cycles")
            torch.cuda.synchronize()
        c = cyc.tolist()
        print(f"lane kernel phases per work item (s_memtime ticks): load+geom+bbox={c[0]/c[4]:.0f} "
              f"stage={c[1]/c[4]:.0f} gather={c[2]/c[4]:.0f} out={c[3]/c[4]:.0f} items={c[4]}; "
              f"all-staged items={c[5]} gather={c[6]/max(c[5],1):.0f} total={c[7]/max(c[5],1):.0f}")
    elif a.phases:
        from egtr_amd import _lib
        cyc = torch.zeros(4, dtype=torch.int64, device=dev)
        out = torch.empty(a.batch, loc.shape[1], 256, device=dev)
        for _ in range(3):
            cyc.zero_()
            st = _lib.lib().egtr_msda_tile_phase_cycles(torch.cuda.current_stream().cuda_stream, value.data_ptr(),
                                                        shp.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
                                                        attn.data_ptr(), a.batch, value.shape[1], shp.shape[0],
                                                        loc.shape[1], 4, out.data_ptr(), cyc.data_ptr())
            _lib.check(st, "phase cycles")
            torch.cuda.synchronize()
        c = cyc.tolist()
        print(f"tile kernel phases per work item (s_memtime ticks): A={c[0]/c[3]:.0f} B={c[1]/c[3]:.0f} "
              f"C={c[2]/c[3]:.0f} items={c[3]}")
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
    alg = B * (min(S * 256 * e, Lq * 8 * 16 * 4 * 32 * e) + Lq * 256 * 4 + Lq * 128 * 4 + Lq * 256 * e)
    if a.bwd:
        alg += B * (Lq * 256 * 4 + 2 * S * 256 * 4 + Lq * 256 * 4 + Lq * 128 * 4)
    print(f"msda {'bwd' if a.bwd else 'fwd'} variant={a.variant} jitter={a.jitter} lq={a.lq} B={B} S={S} Lq={Lq} {'bf16' if a.bf16 else 'f32'}: "
          f"{us:.2f} us/launch (incl. the zero-fill of grad_value for bwd), algorithmic {alg / 1e6:.2f} MB -> "
          f"{alg / us / 1e3:.1f} GB/s = {alg / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
