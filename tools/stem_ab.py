#!/usr/bin/env python3
"""The ResNet stem (7x7/2 convolution + shift + ReLU + 3x3/2 max-pool, channels-last out) at 600x1000: MIOpen convolution + the
pool kernel + the layout change (what the backbone ran) against the one-launch kernel (egtr_stem_conv7x7_pool_x6_f32).
    python tools/stem_ab.py [--batch 1]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv3_fused_ab import graph_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from egtr_amd import ops
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(a.batch, 3, 600, 1000, device=dev)
    w = torch.randn(64, 3, 7, 7, device=dev) / 147 ** 0.5
    b = torch.randn(64, device=dev) * 0.2

    def vendor():
        y = F.conv2d(x, w, None, stride=2, padding=3)
        return ops.bias_relu_maxpool(y, b).contiguous(memory_format=torch.channels_last)

    wxs = ops.stem_weights(w)
    tv, yv = graph_time(vendor)
    tt, yt = graph_time(lambda: ops.stem_fused(x, wxs, b))
    ref = F.max_pool2d(torch.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3) + b.double().view(1, -1, 1, 1)), 3, 2, 1)
    print(f"vendor convolution + pool + layout {tv:6.1f} us (err {float((yv.double() - ref).abs().max()):.1e})   "
          f"one launch {tt:6.1f} us (err {float((yt.double() - ref).abs().max()):.1e})")


if __name__ == "__main__":
    main()
