#!/bin/bash
# Round-6 evidence set (one gpurun call): GPU test suite, kernel stats / forward breakdown / train gaps, counter passes
# (memory: MSDA fwd + relation head + encoder tail + decoder layer + MSDA bwd pair + bf16 MSDA; matrix pipe: x6 kernels + bf16
# matrix kernels), then the default bench command LAST, with the fresh counter summaries already copied to profiles/ so that its
# line quotes them (every summary carries the sha256 of its kernel's sources, tools/kernel_source_hash.py).
# Copy gpurun_out/r06_* to profiles/ afterwards.
set -u
tag=r06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputest.log 2>&1; tail -3 gpurun_out/${tag}_gputest.log
bash tools/profile_r05.sh ${tag} > /dev/null 2>&1
bash tools/profile_r05_all.sh ${tag} > gpurun_out/${tag}_profile_all.log 2>&1
bash tools/stress_mfma_pmc.sh ${tag} > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_stress -o st -- python3 tools/stress_bench.py --iters 4 > gpurun_out/${tag}_stress_run.log 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_stress/st_results.db 24 > gpurun_out/${tag}_stress_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_stress
python3 tools/train_ops.py 400 --shapes > gpurun_out/${tag}_train_ops_by_shape.txt 2>&1
cp gpurun_out/${tag}_*pmc*.json profiles/ 2>/dev/null
( time python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err ) 2> gpurun_out/${tag}_bench_time.txt
cut -c1-900 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err; cat gpurun_out/${tag}_bench_time.txt
ls gpurun_out | grep "^${tag}_" | tr '\n' ' '
