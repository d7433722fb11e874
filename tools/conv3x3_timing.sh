#!/bin/bash
# Build libegtr_hip.so with the 3x3 convolution's phase stamps (-DEGTR_CONV_TIMING) into /tmp and run tools/conv3x3_timing.py.
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/conv3x3_x6.o")
mkdir -p /tmp/ct
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc -DEGTR_CONV_TIMING $1 -c egtr_amd/csrc/conv3x3_x6.hip -o /tmp/ct/t.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ct/t.o -o /tmp/ct/lib.so || exit 1
EGTR_HIP_LIBRARY=/tmp/ct/lib.so timeout 300 python3 tools/conv3x3_timing.py
