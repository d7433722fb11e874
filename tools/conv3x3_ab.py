#!/usr/bin/env python3
"""The 3x3 convolutions of the channels-last fp32 backbone at the 600x1000 shapes: MIOpen (find mode) against the split-bf16
implicit-GEMM kernel (egtr_conv3x3_x6_f32), HIP-graph replayed, 8 calls per graph; error of each against an fp64 convolution.
    python tools/conv3x3_ab.py [--batch 1]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv3_fused_ab import graph_time  # noqa: E402


def variant_exists(C, stride, variant):
    return variant in ({64: (1, 2, 3, 4), 128: (1, 2, 3, 4), 256: (1, 3), 512: (1,)} if stride == 1 else {128: (1,), 256: (1,), 512: (1,)}).get(C, ())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from egtr_amd import ops  # noqa: F401  (sets PYTORCH_MIOPEN_SUGGEST_NHWC)
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for C, H, W, stride, cnt in ((64, 150, 250, 1, 3), (128, 75, 125, 1, 3), (256, 38, 63, 1, 5), (512, 19, 32, 1, 2),
                                 (128, 150, 250, 2, 1), (256, 75, 125, 2, 1), (512, 38, 63, 2, 1)):
        x = torch.randn(a.batch, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5)
        wcl = w.contiguous(memory_format=torch.channels_last)
        tv, yv = graph_time(lambda: F.conv2d(x, wcl, None, stride=stride, padding=1))
        ref = F.conv2d(x.double(), w.double(), None, stride=stride, padding=1)
        line = f"C={C:3d} {H}x{W} stride {stride} (x{cnt}): MIOpen {tv:6.1f} us (err {float((yv.double() - ref).abs().max()):.1e})"
        for variant in (0, 1, 2, 3, 4):
            if not ops.conv3x3_supported(x, C, stride, variant) or (variant and not variant_exists(C, stride, variant)):
                continue
            wxs = ops.conv3x3_weights(w, stride, variant)
            tt, yt = graph_time(lambda: ops.conv3x3(x, wxs, C, stride, variant))
            line += f"   x6 variant {variant}: {tt:6.1f} us (err {float((yt.double() - ref).abs().max()):.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
