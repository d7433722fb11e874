#!/usr/bin/env python3
"""Feasibility probe for the TRAIN step (fp32, bs 4, 600x1000 backbone shapes of layers 2-4): forward + backward of the 3x3
convolutions on NCHW vs channels-last tensors, and of the 1x1 convolutions as F.conv2d (NCHW) vs torch.mm on the channels-last
token matrix.  Graph-free: each op is large enough at bs 4 (hipEvent over 10 iterations)."""
import os
import sys

os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    from egtr_amd.runtime import enable_gemm_tuning
    enable_gemm_tuning()
    dev = torch.device("cuda:0")
    B = 4
    tot = [0.0, 0.0]
    for (C, H, W, s, reps) in ((128, 150, 250, 2, 1), (128, 75, 125, 1, 3), (256, 75, 125, 2, 1), (256, 38, 63, 1, 5),
                               (512, 38, 63, 2, 1), (512, 19, 32, 1, 2)):
        x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
        w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).requires_grad_(True)
        xcl = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wcl = w.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        go = torch.randn_like(F.conv2d(x, w, None, s, 1))
        gocl = go.contiguous(memory_format=torch.channels_last)

        def f_nchw():
            x.grad = w.grad = None
            F.conv2d(x, w, None, s, 1).backward(go)

        def f_cl():
            xcl.grad = wcl.grad = None
            F.conv2d(xcl, wcl, None, s, 1).backward(gocl)

        a, b = t(f_nchw), t(f_cl)
        tot[0] += a * reps
        tot[1] += b * reps
        print(f"3x3 C={C:4d} {H}x{W} s{s} fwd+bwd: NCHW {a:8.1f} us   channels_last {b:8.1f} us   (x{reps} per step)")
    print(f"3x3 total per step: NCHW {tot[0] / 1e3:.2f} ms   channels_last {tot[1] / 1e3:.2f} ms")
    tot = [0.0, 0.0]
    for (Ci, Co, H, W, reps) in ((256, 128, 150, 250, 1), (512, 128, 75, 125, 3), (128, 512, 75, 125, 4), (512, 256, 75, 125, 1),
                                 (1024, 256, 38, 63, 5), (256, 1024, 38, 63, 6), (1024, 512, 38, 63, 1), (2048, 512, 19, 32, 2),
                                 (512, 2048, 19, 32, 3)):
        x = torch.randn(B, Ci, H, W, device=dev, requires_grad=True)
        w = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.05).requires_grad_(True)
        x2 = x.detach().permute(0, 2, 3, 1).reshape(-1, Ci).contiguous().requires_grad_(True)
        w2 = w.detach().view(Co, Ci).clone().requires_grad_(True)
        go = torch.randn(B, Co, H, W, device=dev)
        go2 = go.permute(0, 2, 3, 1).reshape(-1, Co).contiguous()

        def f_conv():
            x.grad = w.grad = None
            F.conv2d(x, w).backward(go)

        def f_mm():
            x2.grad = w2.grad = None
            torch.mm(x2, w2.t()).backward(go2)

        a, b = t(f_conv), t(f_mm)
        tot[0] += a * reps
        tot[1] += b * reps
        print(f"1x1 {Ci:4d}->{Co:4d} {H}x{W} fwd+bwd: conv2d NCHW {a:8.1f} us   mm on tokens {b:8.1f} us   (x{reps} per step)")
    print(f"1x1 total per step: conv2d NCHW {tot[0] / 1e3:.2f} ms   mm on tokens {tot[1] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
