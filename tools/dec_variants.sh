#!/bin/bash
# Build variants of the decoder-layer kernel on the GPU box and run the smallest repro on each (debugging aid).
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v dec_layer.o)
for v in "$@"; do
  mkdir -p /tmp/dv_$v
  flags=""; [ "$v" != "default" ] && flags="-D$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c egtr_amd/csrc/dec_layer.hip -o /tmp/dv_$v/dec_layer.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/dv_$v/dec_layer.o -o /tmp/dv_$v/lib.so || continue
  echo "=== variant $v"
  EGTR_HIP_LIBRARY=/tmp/dv_$v/lib.so timeout 90 python3 tools/dec_repro.py 200 6 time 2>&1 | grep -v "amdgpu.ids\|EGTR_HIP_LIBRARY" | tail -4
done
