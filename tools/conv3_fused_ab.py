#!/usr/bin/env python3
"""The tail of a channels-last fp32 ResNet-50 bottleneck, two ways, per layer of the 600x1000 forward (HIP-graph replayed):
    (a) shift + ReLU pass over conv2's output, vendor fp32 GEMM (conv3), shift + shortcut + ReLU pass   (3 launches)
    (b) ONE launch of the bottleneck-tail kernel (egtr_conv1x1_tail_x6_f32, csrc/conv_tail_x6.hip;
        egtr_amd/backbone.py::Bottleneck.forward_folded_nhwc), with the library's tile choice and with each tile pinned
and the largest difference of each from an fp64 product.  (The same fusion as epilogue flags of the K-staged split-bf16 GEMM,
gemm_split.hip, measured 392 us over the 16 blocks against 479 us for (a) and 274 us for (b): not kept.)
    python tools/conv3_fused_ab.py [--h 600 --w 1000 --batch 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


REPS = 8   # calls per captured graph: a one-kernel graph measures the replay's own launch latency, not the kernel


def graph_time(fn1, iters=30):
    def fn():
        for _ in range(REPS - 1):
            fn1()
        return fn1()

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (iters * REPS), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=600)
    ap.add_argument("--w", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from egtr_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    h, w = (a.h + 3) // 4, (a.w + 3) // 4
    total_a = total_c = 0.0
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), 1):
        if li > 1:
            h, w = (h + 1) // 2, (w + 1) // 2
        M, K, N = a.batch * h * w, planes, 4 * planes
        y2 = torch.randn(M, K, device=dev)
        idt = torch.randn(M, N, device=dev)
        w3 = torch.randn(N, K, device=dev) / K ** 0.5
        b2 = torch.randn(K, device=dev) * 0.1
        b3 = torch.randn(N, device=dev) * 0.1
        src = y2.clone()

        def vendor():
            y = src.clone()              # (the pass is in place on conv2's output; the clone is timed separately below)
            ops.bias_act_rows_(y, b2)
            z = torch.mm(y, w3.t())
            ops.bias_act_rows_(z, b3, idt)
            return z

        def clone_only():
            return src.clone()

        wxs = ops.xs_split(w3, weights=True)

        def tail(tile):
            return lambda: ops.conv1x1_tail(src, b2, wxs, b3, idt, N, tile=tile)

        ta, za = graph_time(vendor)
        tc, _ = graph_time(clone_only)
        sweep = []
        for tile in ((0, 0), (64, 256), (64, 128), (32, 256), (32, 128)):
            if tile[0] == 64 and K > 256:
                continue
            tt, zt = graph_time(tail(tile))
            sweep.append((tile, tt, (zt.double() - torch.relu(torch.relu(y2.double() + b2.double()) @ w3.double().t()
                                                              + b3.double() + idt.double())).abs().max().item()))
        ref = torch.relu(torch.relu(y2.double() + b2.double()) @ w3.double().t() + b3.double() + idt.double())
        ea = (za.double() - ref).abs().max().item()
        print(f"layer{li}: M={M:6d} K={K:3d} N={N:4d}  vendor 3 launches {ta - tc:7.1f} us   (x{blocks} blocks)   "
              f"max err vs fp64: vendor {ea:.2e}", flush=True)
        print("         tail kernel: " + "   ".join(f"{t[0]}x{t[1]} {tt:6.1f} us (err {e:.1e})" for t, tt, e in sweep), flush=True)
        total_a += blocks * (ta - tc)
        total_c += blocks * sweep[0][1]
    print(f"all 16 blocks: vendor {total_a:.1f} us   tail kernel (library's tiles) {total_c:.1f} us")


if __name__ == "__main__":
    main()
