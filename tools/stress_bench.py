#!/usr/bin/env python3
"""Throughput of the BASELINE configs[4] stress shape on ONE GPU: 800x1333, N = 300 queries, 8 decoder layers, bf16,
batch 16 (eager launches and HIP-graph replay).  python tools/stress_bench.py [--batch 16] [--dtype bf16|f32]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--find", type=int, default=1, help="1 = MIOpen find mode (torch.backends.cudnn.benchmark), as bench.py's leg")
    a = ap.parse_args()
    from egtr_amd.runtime import GraphedForward, enable_gemm_tuning
    enable_gemm_tuning()
    if a.find:
        from egtr_amd.runtime import enable_conv_tuning
        enable_conv_tuning()
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev, {"num_queries": 300, "decoder_layers": 8})
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = model.to(dt).eval()
    pv = torch.randn(a.batch, 3, 800, 1333, device=dev, dtype=dt)
    pm = torch.ones(a.batch, 800, 1333, dtype=torch.long, device=dev)
    for graphed in (False, True):
        fwd = GraphedForward(model, enabled=graphed)
        with torch.no_grad():
            for _ in range(3):
                fwd(pv, pm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                fwd(pv, pm)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.iters * 1e3
        print(f"stress shape {a.dtype} batch {a.batch} ({'HIP graph' if graphed and fwd.graphed else 'eager'}): "
              f"{ms:.1f} ms per batch = {a.batch / ms * 1e3:.1f} images/s; peak memory "
              f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
