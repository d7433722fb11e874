#!/bin/bash
# Train-step gap analysis (run on the GPU box through gpurun): untraced hipEvent phases + one kernel trace cut at marker
# kernels -> gpurun_out/<tag>_train_gaps.txt (copy to profiles/).   Usage: bash tools/train_gaps.sh <tag>
set -u
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 tools/train_gaps.py measure gpurun_out/${tag}_train_gaps.json 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_tg -o tg -- python3 tools/train_gaps.py trace > gpurun_out/${tag}_train_gaps_trace.log 2>&1
python3 tools/train_gaps.py report gpurun_out/${tag}_train_gaps.json gpurun_out/prof_tg/tg_results.db > gpurun_out/${tag}_train_gaps.txt 2>&1
rm -rf gpurun_out/prof_tg
head -30 gpurun_out/${tag}_train_gaps.txt | cut -c1-220
