#!/bin/bash
# Build libegtr_hip.so with the bottleneck-tail kernel's phase stamps (-DEGTR_TAIL_TIMING) into /tmp and run tools/tail_timing.py.
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/conv_tail_x6.o")
mkdir -p /tmp/tt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc -DEGTR_TAIL_TIMING $1 -c egtr_amd/csrc/conv_tail_x6.hip -o /tmp/tt/t.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/tt/t.o -o /tmp/tt/lib.so || exit 1
EGTR_HIP_LIBRARY=/tmp/tt/lib.so timeout 300 python3 tools/tail_timing.py
