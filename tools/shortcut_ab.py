#!/usr/bin/env python3
"""The stride-2 shortcut projections (1x1 convolution, stride 2) of ResNet layers 2-4 at the 600x1000 shapes: MIOpen (find mode)
against the one-tap form of the own convolution kernel (egtr_conv1x1_strided_x6_f32); HIP-graph replayed, 8 calls per graph."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv3_fused_ab import graph_time  # noqa: E402


def main():
    from egtr_amd import ops
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for C, N, H, W in ((256, 512, 150, 250), (512, 1024, 75, 125), (1024, 2048, 38, 63)):
        x = torch.randn(1, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(N, C, 1, 1, device=dev) / C ** 0.5
        wcl = w.contiguous(memory_format=torch.channels_last)
        tv, yv = graph_time(lambda: F.conv2d(x, wcl, None, stride=2))
        ref = F.conv2d(x.double(), w.double(), None, stride=2).permute(0, 2, 3, 1).reshape(-1, N)
        wxs = ops.xs_split(w.reshape(N, C).contiguous(), weights=True)
        tt, yt = graph_time(lambda: ops.conv1x1_strided(x, wxs, N, 2))
        ev = float((yv.permute(0, 2, 3, 1).reshape(-1, N).double() - ref).abs().max())
        print(f"C={C} -> {N}, {H}x{W} / 2: MIOpen {tv:6.1f} us (err {ev:.1e})   own {tt:6.1f} us (err {float((yt.double() - ref).abs().max()):.1e})",
              flush=True)


if __name__ == "__main__":
    main()
