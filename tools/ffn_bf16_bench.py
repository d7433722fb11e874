#!/usr/bin/env python3
"""Feed-forward block of the bf16 stress shape (M = 16 x 22 223 rows, 256 -> 1024 -> 256): egtr_ffn_layernorm_bf16 against the
composition it replaces (two vendor GEMMs + the residual / LayerNorm (+ pos) launch).  hipEvent timing."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egtr_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16 * 22223)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fused-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fc1, fc2, ln = torch.nn.Linear(256, 1024), torch.nn.Linear(1024, 256), torch.nn.LayerNorm(256)
    fc1, fc2, ln = fc1.to(dev).bfloat16(), fc2.to(dev).bfloat16(), ln.to(dev).bfloat16()
    x = torch.randn(1, a.rows, 256, device=dev).bfloat16()
    pos = torch.randn(a.rows, 256, device=dev).bfloat16()

    def fused():
        return ops.ffn_layernorm_bf16(x, fc1, fc2, ln, pos)

    def composed():
        h = ops.linear(x, fc1.weight, fc1.bias, relu=True)
        f = ops.linear(h, fc2.weight, fc2.bias)
        return ops.add_layer_norm_pos(f, x, ln, pos)

    with torch.no_grad():
        for name, fn in (("fused egtr_ffn_layernorm_bf16", fused),) + (() if a.fused_only else (("vendor GEMMs + LayerNorm launch", composed),)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:34s}: {e0.elapsed_time(e1) / a.iters * 1e3:8.1f} us per call ({a.rows} rows)")
        import ctypes
        from egtr_amd import _lib
        lib = _lib.lib()
        if hasattr(lib, "egtr_ffn_bf16_stamps"):     # instrumented build (tools/ffn_variants.sh EGTR_FFN_TIMING)
            buf = (ctypes.c_ulonglong * 12)()
            lib.egtr_ffn_bf16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
            lib.egtr_ffn_bf16_stamps(buf, 1)
            fused()
            torch.cuda.synchronize()
            lib.egtr_ffn_bf16_stamps(buf, 0)
            n = max(int(buf[7]), 1)
            names = ["prologue", "fc1 (per launch: sum over tiles)", "bias/relu/pack", "fc2", "wait + barrier", "epilogue", "total",
                     "(workgroups)", "epilogue: residual + sums", "epilogue: LayerNorm -> LDS", "epilogue: rows out"]
            for i, nm in enumerate(names):
                if i == 7:
                    continue
                print(f"   us per call (fused timing {nm:34s}: {buf[i] / n:10.0f} cycles per workgroup (wave 0), {n} workgroups")


if __name__ == "__main__":
    main()
