#!/usr/bin/env python3
"""Smallest run of the cluster decoder against the per-operation decoder (debugging aid; EGTR_HIP_LIBRARY selects a build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import test_gpu_decoder_cluster as T  # noqa: E402

n, layers = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = T._model(n, layers)
pv, pm = T._inputs(1, 96, 128)
ref = T._run(model, pv, pm, fused=False, base=True)
out = T._run(model, pv, pm, fused=True, base=True)
torch.cuda.synchronize()
from egtr_amd import decoder_fused
print("status", decoder_fused.read_status(torch.device("cuda:0")), "max diff",
      max(float((a - b).abs().max()) for a, b in zip(out.decoder_hidden_states, ref.decoder_hidden_states)))

if len(sys.argv) > 3 and sys.argv[3] == "time":
    # average duration of one layer launch: HIP events around every egtr_decoder_layer_f32 call of 30 eager forwards
    from egtr_amd import _lib
    lib = _lib.lib()
    raw = lib.egtr_decoder_layer_f32
    evs = []

    def wrapped(stream, args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = raw(stream, args)
        e1.record()
        evs.append((e0, e1))
        return st

    lib.egtr_decoder_layer_f32 = wrapped
    for _ in range(30):
        T._run(model, pv, pm, fused=True, base=True)
    torch.cuda.synchronize()
    lib.egtr_decoder_layer_f32 = raw
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[len(evs) // 3:])
    print(f"layer launch: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us over {len(ts)} launches (event-bracketed, eager)")
