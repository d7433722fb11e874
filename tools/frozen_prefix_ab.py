#!/usr/bin/env python3
"""The frozen prefix of the training backbone (stem + layer 1, no gradients; bs 4, 600x1000): NCHW folded route (MIOpen
convolutions + shift / shortcut / ReLU passes) against the channels-last inference kernels + one layout change."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import egtr_amd.backbone as bb
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = bb.ResNet50Features().to(dev).train()
    for m in net.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
    for p in list(net.conv1.parameters()) + list(net.layer1.parameters()):
        p.requires_grad_(False)
    x = torch.randn(4, 3, 600, 1000, device=dev)
    params = net._frozen_prefix()
    outs = {}
    for flag in (True, False, True, False):
        bb.FROZEN_PREFIX_NHWC = flag
        net._frozen_folded = None
        for _ in range(3):
            y = net._forward_frozen_prefix(x, params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = net._forward_frozen_prefix(x, params)
        torch.cuda.synchronize()
        print(f"channels-last kernels {int(flag)}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per prefix (bs 4)", flush=True)
        outs[flag] = y
    d = (outs[True] - outs[False]).abs().max().item()
    print(f"max |difference| {d:.2e} on outputs of scale {outs[False].abs().max().item():.1f}; NCHW-contiguous: {outs[True].is_contiguous()}")


if __name__ == "__main__":
    main()
