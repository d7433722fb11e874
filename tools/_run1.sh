#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_fb -o fb -- python bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
python tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 40 > gpurun_out/x_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_fb
