#!/bin/bash
# Final-state profiles of the default bench command (run on the GPU box through gpurun):
#   rocprofv3 --kernel-trace --stats of `python bench.py`, summarised into gpurun_out/ for copying into profiles/.
# Usage: bash tools/profile_bench.sh <tag>      (counters are collected separately: tools/pmc_passes.sh)
set -u
tag=${1:-r01_final}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py > gpurun_out/${tag}_bench_rocprofv3.log 2>&1
grep -E '^\{"metric"' gpurun_out/${tag}_bench_rocprofv3.log > gpurun_out/${tag}_bench.json
python tools/rocpd_stats.py gpurun_out/prof_bench/bench_results.db --top 60 --split-grid msda_fwd_q64:1000 --csv gpurun_out/${tag}_bench_kernel_stats.csv > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_bench
# one graph-replayed forward split into stages (no CPU baseline / parity pass in the trace)
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_fb -o fb -- python bench.py --no-cpu-baseline --extras 0 --steps 20 > /dev/null 2>&1
python tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 14 > gpurun_out/${tag}_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_fb
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/prof_train -o train -- python bench.py --mode train --steps 8 --warmup 6 --no-cpu-baseline --no-kernel-probes > gpurun_out/${tag}_train_rocprofv3.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_train/train_results.db --last-ms 300 --top 60 > gpurun_out/${tag}_train_steady_state_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_train
tail -c 400 gpurun_out/${tag}_bench_rocprofv3.log | head -c 0
cat gpurun_out/${tag}_bench.json | cut -c1-900
grep -E '^\{"metric"' gpurun_out/${tag}_train_rocprofv3.log | cut -c1-330
head -4 gpurun_out/${tag}_train_steady_state_kernel_stats.txt | cut -c1-160
