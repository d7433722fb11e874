#!/bin/bash
# Depth of the weight-fragment queue of the 3x3 convolution kernels (-DEGTR_CONV_PF=n): stand-alone times per layer shape.
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/conv3x3_x6.o")
for v in 3 5 8 11; do
  d=/tmp/cv_$v
  mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc -DEGTR_CONV_PF=$v -c egtr_amd/csrc/conv3x3_x6.hip -o $d/t.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs $d/t.o -o $d/lib.so || continue
  echo "=== EGTR_CONV_PF=$v"
  EGTR_HIP_LIBRARY=$d/lib.so timeout 300 python3 tools/conv3x3_ab.py 2>&1 | grep -E "^C=" | cut -c1-34,85-
done
