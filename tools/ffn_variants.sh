#!/bin/bash
# Ablation builds of csrc/ffn_bf16.hip on the GPU box (debugging aid): bash tools/ffn_variants.sh default EGTR_FFN_ABL_NO_FC2 ...
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v ffn_bf16.o)
for v in "$@"; do
  mkdir -p /tmp/fv_$v
  flags=""; [ "$v" != "default" ] && flags="-D${v//+/ -D}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc $flags -c egtr_amd/csrc/ffn_bf16.hip -o /tmp/fv_$v/ffn_bf16.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fv_$v/ffn_bf16.o -o /tmp/fv_$v/lib.so || continue
  echo "=== variant $v"
  EGTR_HIP_LIBRARY=/tmp/fv_$v/lib.so timeout 120 python3 tools/ffn_bf16_bench.py --fused-only 2>&1 | grep "us per call"
done
