#!/usr/bin/env python3
"""Where do the small ATen launches of one train step come from?  torch.profiler with Python stacks over two steps of the bench's
train workload; device kernels are grouped by the innermost egtr_amd / bench frame of the op that launched them.
    python tools/train_glue_sources.py [--top 40]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    from egtr_amd.runtime import DataParallelTrainer, configure_optimizers
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev, {"dropout": 0.1})
    model.train()
    opt = configure_optimizers(model, lr=2e-6, lr_backbone=2e-7, lr_initialized=None, weight_decay=1e-4)
    tr = DataParallelTrainer(model, optimizer=opt, accumulate=1, clip=0.1, graph=False)
    torch.manual_seed(100)
    b = {"pixel_values": torch.randn(4, 3, bench.H_IMG, bench.W_IMG, device=dev),
         "pixel_mask": torch.ones(4, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev),
         "labels": bench.make_targets(4, cfg, dev, 7)}
    for _ in range(4):
        tr.training_step(b)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    steps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
        for _ in range(steps):
            tr.training_step(b)
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for ev in prof.events():
        if ev.device_type.name == "CUDA" or not ev.kernels:
            continue
        if not ev.name.startswith("aten::") and "Backward" not in ev.name:
            continue
        kn = [k for k in ev.kernels]
        if not kn:
            continue
        site = None
        for fr in (ev.stack or []):
            if ("egtr_amd" in fr or "bench.py" in fr) and "site-packages" not in fr:
                site = fr.replace(root + "/", "")
                break
        if site is None:
            site = "(autograd engine / no python frame): " + ev.name
        rec = by_site[site]
        rec[0] += len(kn)
        rec[1] += sum(k.duration for k in kn)
        rec[2][ev.name] += len(kn)
    rows = sorted(by_site.items(), key=lambda kv: -kv[1][0])
    tot = sum(r[0] for _, r in rows)
    print(f"{tot / steps:.0f} device launches per step from ATen ops, by call site (launches / step, us / step, ops):")
    for site, (n, us, ops_) in rows[:a.top]:
        top = ", ".join(f"{k.replace('aten::', '')}x{v // steps}" for k, v in ops_.most_common(4))
        print(f"{n / steps:7.1f} {us / steps:9.1f}  {site[:110]}  [{top}]")


if __name__ == "__main__":
    main()
