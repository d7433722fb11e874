#!/usr/bin/env python3
"""The FIRST 1x1 convolution of a bottleneck (4 planes -> planes, shift + ReLU) on channels-last fp32 rows at the 600x1000
shapes: vendor GEMM with bias + ReLU epilogue (torch._addmm_activation, what the backbone runs) against the panel-resident
split-bf16 kernel of the bottleneck tail (egtr_conv1x1_tail_x6_f32 without input shift / shortcut), where its shapes allow
(K <= 512, N % 128 == 0).  HIP-graph replayed, 8 calls per graph.
    python tools/conv1_ab.py [--batch 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv3_fused_ab import graph_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from egtr_amd import ops
    from egtr_amd.runtime import enable_gemm_tuning
    enable_gemm_tuning()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    # (rows, K, N, how many such launches per forward)
    shapes = [(37500, 64, 64, 1), (37500, 256, 64, 2), (37500, 256, 128, 1), (9375, 512, 128, 3), (9375, 512, 256, 1),
              (2394, 1024, 256, 5), (2394, 1024, 512, 1), (608, 2048, 512, 2)]
    for M, K, N, cnt in shapes:
        M *= a.batch
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev) * 0.1
        wt = w.t()

        def vendor():
            return torch._addmm_activation(b, x, wt, use_gelu=False)

        for _ in range(3):
            vendor()          # TunableOp picks
        tv, yv = graph_time(vendor)
        line = f"M={M:6d} K={K:4d} N={N:3d} (x{cnt}): vendor {tv:6.1f} us"
        if ops.conv1x1_tail_supported(x, N):
            wxs = ops.xs_split(w, weights=True)
            for tile in ((0, 0),) + (((32, 128), (64, 128)) if N % 128 == 0 else ()) + (((32, 256),) if N % 256 == 0 else ()):
                if tile[0] == 64 and K > 256:
                    continue
                tt, yt = graph_time(lambda: ops.conv1x1_tail(x, None, wxs, b, None, N, relu_in=False, relu_out=True, tile=tile))
                err = (yt.double() - torch.relu(x.double() @ w.double().t() + b.double())).abs().max().item()
                line += f"   tail {tile[0]}x{tile[1]} {tt:6.1f} us (err {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
