#!/bin/bash
# Ablation builds of csrc/rel_head_bf16.hip on the GPU box (debugging aid): bash tools/rel_head_variants.sh default EGTR_RH_ABL_NO_L1 ...
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v rel_head_bf16.o)
for v in "$@"; do
  mkdir -p /tmp/rv_$v
  flags=""; [ "$v" != "default" ] && flags="-D${v//+/ -D}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc $flags -c egtr_amd/csrc/rel_head_bf16.hip -o /tmp/rv_$v/rel_head_bf16.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/rv_$v/rel_head_bf16.o -o /tmp/rv_$v/lib.so || continue
  echo "=== variant $v"
  EGTR_HIP_LIBRARY=/tmp/rv_$v/lib.so timeout 120 python3 tools/rel_head_bf16_bench.py 2>&1 | grep "packed"
done
