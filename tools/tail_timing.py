#!/usr/bin/env python3
"""Where a workgroup of the fp32 bottleneck-tail kernel spends its time (library built with -DEGTR_TAIL_TIMING, see
tools/tail_timing.sh): per layer shape of the 600x1000 forward, average shader-clock cycles of wave 0 per phase, the average
lifetime of a workgroup and the span from the first workgroup's start to the last one's end on the 100 MHz real-time counter."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from egtr_amd import ops, _lib
    lib = ctypes.CDLL(os.environ["EGTR_HIP_LIBRARY"])
    lib.egtr_conv_tail_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    import numpy as np
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    for li, (M, K, N) in enumerate(((37500, 64, 256), (9375, 128, 512), (2394, 256, 1024), (608, 512, 2048)), 1):
        a = torch.randn(M, K, device=dev)
        sc = torch.randn(M, N, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b2, b3 = torch.randn(K, device=dev), torch.randn(N, device=dev)
        wxs = ops.xs_split(w, weights=True)
        junk = torch.empty(64 * 1024 * 1024, device=dev)
        for tile in ((0, 0), (32, 256)):
            for _ in range(3):
                ops.conv1x1_tail(a, b2, wxs, b3, sc, N, tile=tile)
            junk.zero_()          # push the operands out of the L2s, as the rest of the forward does
            torch.cuda.synchronize()
            lib.egtr_conv_tail_stamps(buf, 1)
            ops.conv1x1_tail(a, b2, wxs, b3, sc, N, tile=tile)
            torch.cuda.synchronize()
            lib.egtr_conv_tail_stamps(buf, 0)
            r = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
            r = r[r[:, 7] == 1]
            t0 = r[:, 0].min()
            start, end = (r[:, 0] - t0) * 10, (r[:, 1] - t0) * 10
            ph = r[:, 2:7].mean(axis=0) / 2.25   # shader clock ~2.25 GHz (lifetime / phase sum of these runs) -> ns
            q = lambda x: "/".join(f"{int(v)}" for v in np.percentile(x, [0, 25, 50, 75, 100]))  # noqa: E731
            print(f"layer{li} tile {tile[0]}x{tile[1]}: {len(r)} workgroups, span {end.max()} ns;  per workgroup (ns): requests + panel "
                  f"{ph[0]:6.0f}  barrier {ph[1]:5.0f}  products {ph[2]:6.0f}  epilogue {ph[3]:6.0f}  store drain {ph[4]:6.0f};  "
                  f"lifetime {np.mean(end - start):6.0f}", flush=True)
            print(f"         start times (min/25/50/75/max) {q(start)}   end times {q(end)}", flush=True)


if __name__ == "__main__":
    main()
