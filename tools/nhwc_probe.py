#!/usr/bin/env python3
"""Feasibility probe (bf16, stress shapes, bs 16): MIOpen 3x3 convolutions on NCHW vs channels_last tensors
(PYTORCH_MIOPEN_SUGGEST_NHWC=1), and 1x1 convolutions as an NCHW batched GEMM (backbone.conv1x1_as_gemm) vs a plain
[B*H*W, Cin] x [Cin, Cout] GEMM with bias + ReLU in the epilogue on the channels_last tensor."""
import os
import sys
import time

os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def t(fn, n=20):
    """us per call; the calls are replayed from a HIP graph (20 per replay) so that small launches are not measured at the CPU's
    launch rate"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


def main():
    from egtr_amd.runtime import enable_gemm_tuning
    from egtr_amd.backbone import conv1x1_as_gemm
    enable_gemm_tuning()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    fp32 = "--fp32" in sys.argv       # the headline workload: fp32, bs 1, 600x1000
    B = 1 if fp32 else 16
    dt = torch.float32 if fp32 else torch.bfloat16
    s3 = (((64, 150, 250, 1), (128, 150, 250, 2), (128, 75, 125, 1), (256, 75, 125, 2), (256, 38, 63, 1), (512, 38, 63, 2),
           (512, 19, 32, 1)) if fp32 else
          ((64, 200, 334, 1), (128, 200, 334, 2), (128, 100, 167, 1), (256, 100, 167, 2), (256, 50, 84, 1),
           (512, 50, 84, 2), (512, 25, 42, 1)))
    for (C, H, W, s) in s3:
        x = torch.randn(B, C, H, W, device=dev, dtype=dt)
        w = torch.randn(C, C, 3, 3, device=dev, dtype=dt) * 0.05
        xcl, wcl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        a = t(lambda: F.conv2d(x, w, None, s, 1))
        b = t(lambda: F.conv2d(xcl, wcl, None, s, 1))
        y = F.conv2d(xcl, wcl, None, s, 1)
        print(f"3x3 C={C:4d} {H}x{W} s{s}: NCHW {a:8.1f} us   channels_last {b:8.1f} us   (output channels_last: "
              f"{y.is_contiguous(memory_format=torch.channels_last)})")
    s1 = (((64, 256, 150, 250), (256, 64, 150, 250), (512, 128, 75, 125), (128, 512, 75, 125), (1024, 256, 38, 63),
           (256, 1024, 38, 63), (2048, 512, 19, 32), (512, 2048, 19, 32)) if fp32 else
          ((64, 256, 200, 334), (256, 64, 200, 334), (512, 128, 100, 167), (128, 512, 100, 167),
           (1024, 256, 50, 84), (256, 1024, 50, 84), (2048, 512, 25, 42), (512, 2048, 25, 42)))
    for (Ci, Co, H, W) in s1:
        x = torch.randn(B, Ci, H, W, device=dev, dtype=dt)
        w = torch.randn(Co, Ci, 1, 1, device=dev, dtype=dt) * 0.05
        bias = torch.randn(Co, device=dev, dtype=dt)
        xcl = x.contiguous(memory_format=torch.channels_last)
        x2 = xcl.permute(0, 2, 3, 1).reshape(-1, Ci)
        w2 = w.view(Co, Ci)
        a = t(lambda: conv1x1_as_gemm(x, w))
        b = t(lambda: torch._addmm_activation(bias, x2, w2.t(), use_gelu=False))
        c = t(lambda: torch.addmm(bias, x2, w2.t()))
        print(f"1x1 {Ci:4d}->{Co:4d} {H}x{W}: NCHW GEMM (no epilogue) {a:8.1f} us   NHWC addmm+ReLU {b:8.1f} us   NHWC addmm {c:8.1f} us")


if __name__ == "__main__":
    main()
