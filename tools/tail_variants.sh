#!/bin/bash
# Variants of the fp32 bottleneck-tail kernel for the long-K layers (conv_tail_x6.hip): depth of the weight-fragment queue
# (-DEGTR_TAIL_PF_LONG=n) and 256-column blocks as 8 waves x 32 columns (-DEGTR_TAIL_NW8_LONGK).  Stand-alone times per layer.
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/conv_tail_x6.o")
for v in default "EGTR_TAIL_PF_LONG=5" "EGTR_TAIL_PF_LONG=8" "EGTR_TAIL_NW8_LONGK" "EGTR_TAIL_NW8_LONGK -DEGTR_TAIL_PF_LONG=6"; do
  d=/tmp/tv_$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  mkdir -p $d
  flags=""; [ "$v" != "default" ] && flags="-D$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc $flags -c egtr_amd/csrc/conv_tail_x6.hip -o $d/t.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs $d/t.o -o $d/lib.so || continue
  echo "=== variant $v"
  EGTR_HIP_LIBRARY=$d/lib.so timeout 300 python3 tools/conv3_fused_ab.py 2>&1 | grep -E "tail kernel|all 16"
done
