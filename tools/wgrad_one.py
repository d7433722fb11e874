#!/usr/bin/env python3
"""One shape of the training-side kernels under a profiler / PMC pass: weight gradient g^T x (split-bf16, split-K), bias
gradient (column sum) and data gradient of a token-sized linear layer.   python tools/wgrad_one.py [M] [N] [K] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from egtr_amd import ops
    a = [int(v) for v in sys.argv[1:]]
    M, N, K, reps = (a + [50148, 256, 256, 20][len(a):])[:4]
    g = torch.randn(M, N, device="cuda")
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    wt_t = ops.gemm_split_tile(w, transposed=True) if K % 128 == 0 and N % 32 == 0 else None
    for _ in range(reps):
        ops.linear_split_bf16_wgrad(g, x)
        ops.column_sum(g)
        if wt_t is not None:
            ops.linear_split_bf16(g, wt_t, None, K)
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: {reps} x (weight gradient, column sum, data gradient)")


if __name__ == "__main__":
    main()
