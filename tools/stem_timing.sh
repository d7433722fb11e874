#!/bin/bash
# Where a workgroup of the fused stem kernel spends its time (-DEGTR_STEM_TIMING: per-workgroup phase stamps of one launch).
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/stem_x6.o")
mkdir -p /tmp/st
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc -DEGTR_STEM_TIMING -c egtr_amd/csrc/stem_x6.hip -o /tmp/st/t.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/st/t.o -o /tmp/st/lib.so || exit 1
EGTR_HIP_LIBRARY=/tmp/st/lib.so timeout 300 python3 - <<'PY'
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from egtr_amd import ops
lib = ctypes.CDLL(os.environ["EGTR_HIP_LIBRARY"])
lib.egtr_stem_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(1, 3, 600, 1000, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) / 12
b = torch.randn(64, device=dev)
wx = ops.stem_weights(w)
junk = torch.empty(64 * 1024 * 1024, device=dev)
for _ in range(3):
    ops.stem_fused(x, wx, b)
junk.zero_()
torch.cuda.synchronize()
ops.stem_fused(x, wx, b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4096 * 8))()
lib.egtr_stem_stamps(buf)
r = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
r = r[r[:, 7] == 1]
t0 = r[:, 0].min()
start, end = (r[:, 0] - t0) * 10, (r[:, 1] - t0) * 10
life = np.mean(end - start)
ph = r[:, 2:7].mean(axis=0)
ph = ph / ph.sum() * life
print(f"{len(r)} workgroups, span {end.max()} ns; per workgroup (ns): input tile {ph[0]:.0f}  barrier {ph[1]:.0f}  products + LDS {ph[2]:.0f}  "
      f"barrier {ph[3]:.0f}  pool + stores {ph[4]:.0f}; lifetime {life:.0f}; starts {np.percentile(start, [0, 50, 100])}  ends {np.percentile(end, [0, 50, 100])}")
PY
