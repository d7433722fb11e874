#!/usr/bin/env python3
"""Where does the HOST time of one eager bs = 1 forward go?  cProfile over eager forwards of the bench model (no graph replay):
the eager forward is host-bound (4.3 ms against 3.1 ms of kernels), i.e. every Python-side microsecond per launch shows.
    python tools/eager_host_profile.py [n_forwards] [rows]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    dev = torch.device("cuda:0")
    from egtr_amd.runtime import enable_conv_tuning, enable_gemm_tuning
    enable_conv_tuning()
    enable_gemm_tuning()
    model, cfg, _ = bench.build_model(dev)
    pv = torch.randn(1, 3, bench.H_IMG, bench.W_IMG, device=dev)
    pm = torch.ones(1, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev)

    def fwd():
        return model(pixel_values=pv, pixel_mask=pm, output_attentions=False, output_attention_states=True,
                     output_hidden_states=True)

    with torch.no_grad():
        for _ in range(10):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fwd()
        t_host = time.perf_counter() - t0            # host time to ENQUEUE n forwards (no synchronisation inside)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{n} eager forwards: host enqueue {t_host / n * 1e3:.3f} ms per forward, wall {t_all / n * 1e3:.3f} ms per forward")
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            fwd()
        pr.disable()
        torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(rows)


if __name__ == "__main__":
    main()
