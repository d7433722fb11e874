#!/usr/bin/env python3
"""Where a workgroup of the 3x3 convolution kernel spends its time (library built with -DEGTR_CONV_TIMING, tools/conv3x3_timing.sh):
per-workgroup phase stamps of one launch with cold L2s, as inside the forward."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from egtr_amd import ops
    lib = ctypes.CDLL(os.environ["EGTR_HIP_LIBRARY"])
    lib.egtr_conv3x3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    junk = torch.empty(64 * 1024 * 1024, device=dev)
    for C, H, W, variants in ((64, 150, 250, (1, 4)), (128, 75, 125, (0, 4)), (256, 38, 63, (0,))):
        x = torch.randn(1, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
        for variant in variants:
            wxs = ops.conv3x3_weights(w, 1, variant)
            for cold in (True,):
                for _ in range(3):
                    ops.conv3x3(x, wxs, C, 1, variant)
                if cold:
                    junk.zero_()
                torch.cuda.synchronize()
                lib.egtr_conv3x3_stamps(buf, 1)
                ops.conv3x3(x, wxs, C, 1, variant)
                torch.cuda.synchronize()
                lib.egtr_conv3x3_stamps(buf, 0)
                r = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
                r = r[r[:, 7] == 1]
                t0 = r[:, 0].min()
                start, end = (r[:, 0] - t0) * 10, (r[:, 1] - t0) * 10
                life = np.mean(end - start)
                ph = r[:, 2:7].mean(axis=0)
                ph = ph / ph.sum() * life          # cycles -> ns by the workgroups' own lifetime
                q = lambda v: "/".join(f"{int(z)}" for z in np.percentile(v, [0, 50, 100]))  # noqa: E731
                print(f"C={C} variant {variant} {'cold' if cold else 'warm'}: {len(r)} workgroups, span {end.max()} ns; per workgroup (ns): "
                      f"halo {ph[0]:6.0f}  barrier {ph[1]:5.0f}  products {ph[2]:6.0f}  epilogue {ph[3]:5.0f}  drain {ph[4]:5.0f}; "
                      f"lifetime {life:6.0f}; starts {q(start)}  ends {q(end)}", flush=True)


if __name__ == "__main__":
    main()
