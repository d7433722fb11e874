#!/usr/bin/env python3
"""Per-phase shader-clock cycles of the MSDA region kernel's window scheme (needs a library built with
-DEGTR_REGION_PROF: `make -C egtr_amd/csrc clean all EXTRA=-DEGTR_REGION_PROF`).  Thread 0 of every workgroup accumulates."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import msda_bench as MB
from egtr_amd.load_custom import load_hip_kernels
from egtr_amd import _lib
k = load_hip_kernels()
value, shp, lsi, loc, attn = MB.make_inputs(1, "enc", float(sys.argv[1]) if len(sys.argv) > 1 else 0.0, "cuda:0")
h = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
for _ in range(3):
    k.ms_deform_attn_forward_variant(value, shp, lsi, loc, attn, 15)
torch.cuda.synchronize()
h.egtr_debug_region_prof(buf)
n = 20
for _ in range(n):
    k.ms_deform_attn_forward_variant(value, shp, lsi, loc, attn, 15)
torch.cuda.synchronize()
h.egtr_debug_region_prof(buf)
wg = max(buf[15], 1)
names = ["prologue+probe", "operands+softmax", "barrier (x4)", "next loads+geometry (x4)", "gather (x4)",
         "next window stores (x4)"] + [""] * 8 + ["out stores"]
tot = 0
for i, nm in enumerate(names):
    if not nm:
        continue
    slot = i
    c = buf[slot] / wg
    tot += c
    print(f"{nm:28s} {c:9.0f} ticks per workgroup")
print(f"{'sum':28s} {tot:9.0f} ticks (s_memtime runs at 100 MHz: x 10 ns)")
print("slow-path wave entries per level:", [buf[16 + s] / n for s in range(4)], "outlier samples per level:",
      [buf[20 + s] / n for s in range(4)], "unstaged wave entries:", [buf[24 + s] / n for s in range(4)])
