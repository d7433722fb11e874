#!/bin/bash
# A/B of the bf16 MSDA forward's inner product: v_dot2_f32_bf16 on the packed word with the corner weight rounded to bf16
# (default) vs v_pk_fma_f32 on unpacked pairs (-DEGTR_MSDA_BF16_PKFMA).  Whole stress forward (tools/stress_bench.py) and the plain bf16 entry at the stress shape.
cd "$GRAFT_REPO_ROOT"
objs=$(ls egtr_amd/csrc/*.o | grep -v "csrc/msda.o")
for v in default EGTR_MSDA_BF16_PKFMA; do
  mkdir -p /tmp/md_$v
  flags=""; [ "$v" != "default" ] && flags="-D$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iegtr_amd/csrc $flags -c egtr_amd/csrc/msda.hip -o /tmp/md_$v/msda.o || continue
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/md_$v/msda.o -o /tmp/md_$v/lib.so || continue
  echo "=== variant $v"
  EGTR_HIP_LIBRARY=/tmp/md_$v/lib.so timeout 300 python3 tools/stress_bench.py --iters 8 2>&1 | grep "HIP graph"
  EGTR_HIP_LIBRARY=/tmp/md_$v/lib.so timeout 300 python3 tools/msda_bench.py --big --bf16 --batch 16 --iters 50 2>&1 | grep "us/launch"
  EGTR_HIP_LIBRARY=/tmp/md_$v/lib.so timeout 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "msda_bf16_forward or msda_fused_bf16" 2>&1 | tail -1
done
