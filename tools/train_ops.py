#!/usr/bin/env python3
"""torch.profiler table of a few steady-state train steps of the bench model: which ATen ops / kernels the step spends its
GPU time and its launches on.   python tools/train_ops.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from egtr_amd.runtime import DataParallelTrainer, configure_optimizers, enable_gemm_tuning
    shapes = "--shapes" in sys.argv          # group by input shapes as well (which elementwise passes are token-sized)
    argv = [a for a in sys.argv[1:] if a != "--shapes"]
    rows = int(argv[0]) if argv else 45
    enable_gemm_tuning()
    dev = torch.device("cuda", 0)
    model, cfg, cfg_dict = bench.build_model(dev, {"dropout": 0.1})
    model.train()
    opt = configure_optimizers(model, lr=2e-6, lr_backbone=2e-7, lr_initialized=None, weight_decay=1e-4)
    tr = DataParallelTrainer(model, optimizer=opt, accumulate=1, clip=0.1)
    torch.manual_seed(100)
    batch = {"pixel_values": torch.randn(4, 3, bench.H_IMG, bench.W_IMG, device=dev),
             "pixel_mask": torch.ones(4, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev),
             "labels": bench.make_targets(4, cfg, dev, 7)}
    for _ in range(5):
        tr.training_step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=shapes) as prof:
        for _ in range(3):
            tr.training_step(batch)
        torch.cuda.synchronize()
    if shapes:
        evs = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
        evs.sort(key=lambda e: -e.self_device_time_total)
        print("self GPU ms / 3 steps   calls   op   input shapes")
        for e in evs[:rows]:
            print(f"{e.self_device_time_total / 1e3:9.3f} {e.count:6d}  {e.key[:44]:44s} {str(e.input_shapes)[:110]}")
        return
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=rows, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
