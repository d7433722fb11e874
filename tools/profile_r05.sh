#!/bin/bash
# Round-5 profile of the default bench command: kernel stats + forward breakdown (copy the summaries to profiles/).
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_fb -o fb -- python3 bench.py --no-cpu-baseline --extras 0 --steps 20 > gpurun_out/${tag}_fb_bench.log 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 14 > gpurun_out/${tag}_forward_breakdown.txt 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 14 seq > gpurun_out/${tag}_forward_sequence.txt 2>&1
python3 tools/rocpd_stats.py gpurun_out/prof_fb/fb_results.db --top 40 > gpurun_out/${tag}_fb_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_fb
grep -E '^\{"metric"' gpurun_out/${tag}_fb_bench.log | cut -c1-400
cat gpurun_out/${tag}_forward_breakdown.txt | cut -c1-170
