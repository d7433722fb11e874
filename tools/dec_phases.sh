#!/bin/bash
# Instrumented build of the decoder-layer kernel + its phase table (gpurun; output: gpurun_out/<tag>_dec_phases.txt).
set -u
tag=${1:-r05}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out /tmp/dect
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DEGTR_DEC_TIMING -c egtr_amd/csrc/dec_layer.hip -o /tmp/dect/dec_layer.o || exit 1
objs=$(ls egtr_amd/csrc/*.o | grep -v dec_layer.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/dect/dec_layer.o -o /tmp/dect/libegtr_timing.so || exit 1
EGTR_HIP_LIBRARY=/tmp/dect/libegtr_timing.so python3 tools/dec_phases.py > gpurun_out/${tag}_dec_phases.txt 2>gpurun_out/${tag}_dec_phases.err
cat gpurun_out/${tag}_dec_phases.txt
tail -3 gpurun_out/${tag}_dec_phases.err
