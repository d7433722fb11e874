#!/usr/bin/env python3
"""Host ENQUEUE time of one eager bs = 1 forward by stage (backbone / encoder / decoder / heads), perf_counter around the stage
entry points, no synchronisation inside: the eager forward is host-bound (3.46 ms of enqueue against 3.1 ms of kernels), and the
decoder + heads stages cost 0.88 ms of host time for 14 launches.   python tools/eager_host_stages.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from egtr_amd.runtime import enable_conv_tuning, enable_gemm_tuning
enable_conv_tuning(); enable_gemm_tuning()
dev = torch.device("cuda:0")
model, cfg, _ = bench.build_model(dev)
pv = torch.randn(1, 3, 600, 1000, device=dev); pm = torch.ones(1, 600, 1000, dtype=torch.long, device=dev)
m = model.model
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(m.backbone.conv_encoder.model, "forward", "backbone")
wrap(m.encoder, "forward", "encoder")
wrap(m.decoder, "forward", "decoder")
wrap(model, "_heads", "heads")
wrap(m, "forward", "base_model_total")
with torch.no_grad():
    for _ in range(10): model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
    torch.cuda.synchronize(); acc.clear()
    n = 100; t0 = time.perf_counter()
    for _ in range(n): model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
    tt = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"host enqueue per forward: {tt/n*1e3:.3f} ms")
for k, v in acc.items(): print(f"  {k:20s} {v/n*1e3:.3f} ms")
