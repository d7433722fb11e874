#!/usr/bin/env python3
"""bench.py's throughput-mode leg (bs 8, fp32, 600x1000) with the bottleneck-tail kernel on / off and MIOpen find mode on / off.
    python tools/batched_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import egtr_amd.backbone as bb
    from egtr_amd.runtime import enable_gemm_tuning
    enable_gemm_tuning()
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev, {})
    model = model.eval()
    for find in (False, True):
        for fused in (True, False, True, False):
            bb.CONV3_FUSED = fused
            r = bench.batched_leg(model, dev, find=find)
            print(f"find {int(find)}  tail kernel {int(fused)}: {r['value']} images/s  {r['ms_per_step']} ms per batch of 8", flush=True)


if __name__ == "__main__":
    main()
