#!/bin/bash
# Kernel list of one graph-replayed bs-8 fp32 forward (bench.py's throughput-mode leg), heuristic picks and find mode.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for find in 0 1; do
  rm -rf gpurun_out/prof_b8
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_b8 -o b8 -- python3 -c "
import sys, torch
sys.path.insert(0, '.')
import bench
from egtr_amd.runtime import enable_gemm_tuning
enable_gemm_tuning()
dev = torch.device('cuda:0')
model, cfg, _ = bench.build_model(dev, {})
r = bench.batched_leg(model.eval(), dev, find=bool($find))
print(r['value'], r['ms_per_step'])
" 2>&1 | grep -v -E "amdgpu.ids|rocprofv3|simple_timer" | tail -1
  python3 tools/forward_breakdown.py gpurun_out/prof_b8/b8_results.db 14 > gpurun_out/r06_batched_bs8_breakdown_find$find.txt 2>&1
  rm -rf gpurun_out/prof_b8
  sed -n 1,20p gpurun_out/r06_batched_bs8_breakdown_find$find.txt | cut -c1-150
done
