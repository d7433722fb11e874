#!/usr/bin/env python3
"""Per-stage GPU time of one bench forward (eager launches, HIP events): backbone / input_proj+flatten / encoder /
decoder / heads+relation head.  Diagnostic for DESIGN.md section 4.5."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev)
    pv = torch.randn(1, 3, bench.H_IMG, bench.W_IMG, device=dev)
    pm = torch.ones(1, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev)
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    m = model.model
    hooks = []
    hooks.append(m.backbone.register_forward_pre_hook(lambda *a: mark("start")))
    hooks.append(m.backbone.register_forward_hook(lambda *a: mark("backbone")))
    hooks.append(m.encoder.register_forward_pre_hook(lambda *a: mark("input_proj+flatten")))
    hooks.append(m.encoder.register_forward_hook(lambda *a: mark("encoder")))
    hooks.append(m.decoder.register_forward_hook(lambda *a: mark("decoder")))
    tot = {}
    with torch.no_grad():
        for it in range(8):
            marks.clear()
            out = model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
            mark("heads+relation_head")
            torch.cuda.synchronize()
            if it >= 3:
                for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                    tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
    n = 5
    s = 0.0
    for k, v in tot.items():
        print(f"{k:24s} {v / n:8.3f} ms")
        s += v / n
    print(f"{'total (eager)':24s} {s:8.3f} ms")


if __name__ == "__main__":
    main()
