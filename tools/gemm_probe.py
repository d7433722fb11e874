#!/usr/bin/env python3
"""Time the encoder-shaped fp32 GEMMs (M = 12537 tokens) under the available torch/rocBLAS/hipBLASLt paths."""
import os
import sys
import time

import torch

M = 12537
shapes = [(256, 256), (256, 384), (256, 1024), (1024, 256)]
dev = "cuda:0"


def bench(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    print("tunableop:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"), "blas pref:", torch.backends.cuda.preferred_blas_library())
    for lib in ("default", "hipblaslt", "hipblas"):
        if lib != "default":
            try:
                torch.backends.cuda.preferred_blas_library(lib)
            except Exception as e:
                print(lib, "unavailable", e)
                continue
        for (K, N) in shapes:
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev)
            b = torch.randn(N, device=dev)
            us = bench(lambda: torch.nn.functional.linear(x, w, b))
            us2 = bench(lambda: torch.addmm(b, x, w.t()))
            us3 = bench(lambda: torch.mm(x, w.t()))
            fl = 2.0 * M * K * N
            print(f"{lib:10s} K={K:5d} N={N:5d}: linear {us:7.1f} us ({fl / us / 1e6:6.1f} TF)  addmm {us2:7.1f}  mm {us3:7.1f} ({fl / us3 / 1e6:6.1f} TF)")


if __name__ == "__main__":
    main()
