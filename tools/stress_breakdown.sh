#!/bin/bash
# Full per-kernel list of one graph-replayed bf16 stress forward (800x1333, N = 300, 8 decoder layers, bs 16).
set -u
tag=${1:-r05b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_stress -o st -- python3 tools/stress_bench.py --iters 4 > gpurun_out/${tag}_stress_run.log 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_stress/st_results.db 200 > gpurun_out/${tag}_stress_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_stress
tail -5 gpurun_out/${tag}_stress_run.log
