#!/usr/bin/env python3
"""Token-sized linear layers of the encoder: split-bf16 GEMM (csrc/gemm_split.hip) vs the vendor fp32 GEMM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters=20, reps=10):
    """Average duration of one call, replayed from a HIP graph of `iters` back-to-back launches (no CPU launch floor)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


def column_sum(g):
    from egtr_amd import _lib, ops
    lib = _lib.lib()
    M, N = g.shape
    ws = torch.empty(int(lib.egtr_column_sum_workspace_floats(M, N)), dtype=torch.float32, device=g.device)
    out = torch.empty(N, dtype=torch.float32, device=g.device)
    _lib.check(lib.egtr_column_sum_f32(ops._stream(), g.data_ptr(), None, None, ws.data_ptr(), out.data_ptr(), M, N),
               "egtr_column_sum_f32")
    return out


def main():
    if os.environ.get("EGTR_LIB"):          # A/B of kernel builds: another build of libegtr_hip.so
        from egtr_amd import _lib
        _lib.LIB_PATH = os.path.abspath(os.environ["EGTR_LIB"])
    from egtr_amd import ops, runtime
    if len(sys.argv) > 1 and sys.argv[1] == "tune":
        runtime.enable_gemm_tuning()
    M = int(os.environ.get("GEMM_BENCH_ROWS", "12537"))   # 50148 = the training batch (4 images)
    for K, N, relu in ((256, 256, False), (256, 384, False), (256, 1024, True), (1024, 256, False), (384, 256, False)):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        wt = ops.gemm_split_weights(w)
        with torch.no_grad():
            t_split = timeit(lambda: ops.linear_split_bf16(x, wt, b, N, relu=relu))
            t_vendor = timeit(lambda: ops.linear(x, w, b, 1.0, relu))
            t_mm = timeit(lambda: x.mm(w.t()))
            g = torch.randn(M, N, device="cuda")
            t_wgrad = timeit(lambda: x.t().mm(g))
            t_wsplit = timeit(lambda: ops.linear_split_bf16_wgrad(g, x)) if N % 128 == 0 and K % 128 == 0 else float("nan")
            t_sum = timeit(lambda: g.sum(0))
            t_col = timeit(lambda: column_sum(g))
        fl = 2.0 * M * K * N
        print(f"    mm without bias {t_mm:.1f} us, weight gradient x^T g {t_wgrad:.1f} us (split-bf16 {t_wsplit:.1f} us); column sum of [M, {N}]: "
              f"torch {t_sum:.1f} us, egtr_column_sum_f32 {t_col:.1f} us")
        print(f"M={M} K={K} N={N} relu={relu}: split-bf16 {t_split:.1f} us ({fl / t_split / 1e6:.0f} TFLOP/s), "
              f"vendor fp32 {t_vendor:.1f} us ({fl / t_vendor / 1e6:.0f} TFLOP/s)")


if __name__ == "__main__":
    main()
