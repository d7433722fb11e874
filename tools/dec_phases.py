#!/usr/bin/env python3
"""Per-phase time stamps of the decoder-layer cluster kernel (csrc/dec_layer.hip built with -DEGTR_DEC_TIMING, see
tools/dec_phases.sh): the eight workgroups of cluster 0, 100 MHz wall clock, one table per layer of one eager forward of
the bench model (600x1000, N = 200)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NAMES = ["start", "scores", "softmax", "P.V", "out-proj+store", "barrier 1", "reduce+LN1", "off/logit gemm", "records",
         "gather", "cross-proj+store", "barrier 2", "reduce+LN2", "fc1", "hidden", "fc2+store", "barrier 3", "reduce+LN3",
         "qkv+store"]


def main():
    from egtr_amd import _lib
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev)
    pv = torch.randn(1, 3, bench.H_IMG, bench.W_IMG, device=dev)
    pm = torch.ones(1, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev)
    lib = _lib.lib()
    raw = lib.egtr_decoder_layer_f32
    stamps_fn = ctypes.CDLL(_lib.LIB_PATH).egtr_decoder_layer_stamps
    stamps_fn.argtypes = [ctypes.c_void_p]
    tables = []

    def wrapped(stream, args):
        st = raw(stream, args)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (8 * 32))()
        assert stamps_fn(buf) == 0
        tables.append([[buf[h * 32 + i] for i in range(19)] for h in range(8)])
        return st

    with torch.no_grad():
        for _ in range(3):
            model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
        lib.egtr_decoder_layer_f32 = wrapped
        model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
        lib.egtr_decoder_layer_f32 = raw
    for li, t in enumerate(tables):
        t0 = min(row[0] for row in t)
        print(f"layer {li}: microseconds since the first workgroup of cluster 0 started (columns: heads 0..7), then the "
              f"slowest head's phase duration")
        for i in range(19):
            if all(row[i] == 0 for row in t):
                continue
            cells = " ".join(f"{(row[i] - t0) / 100.0:7.2f}" for row in t)
            dur = max((row[i] - row[i - 1]) / 100.0 for row in t) if i else 0.0
            print(f"  {NAMES[i]:18s} {cells}   | {dur:6.2f}")
        end = max(max(row[:19]) for row in t)
        print(f"  cluster 0 done after {(end - t0) / 100.0:.2f} us")


if __name__ == "__main__":
    main()
