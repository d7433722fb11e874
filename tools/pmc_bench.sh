#!/bin/bash
# PMC passes (memory counters) over the default bench command, summarised into gpurun_out/<tag>_msda_pmc.json
# (copy to profiles/ after review).  Usage: bash tools/pmc_bench.sh <tag> <kernel-regex> <kernel-label>
set -u
tag=${1:-r02}; rx=${2:-'msda_fwd_q64_f32<true, false'}; label=${3:-'msda_fwd_q64_f32<fused prologue>'}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/pmc_passes.sh gpurun_out/pmc_${tag} bench mem -- python3 bench.py --no-cpu-baseline --extras 0 --steps 5 --warmup 2 --tune-gemm 0
python3 tools/msda_pmc.py gpurun_out/pmc_${tag} --kernel-regex "$rx" --name "$label" --alg-bytes 44932608 \
    --out gpurun_out/${tag}_msda_pmc.json > gpurun_out/${tag}_msda_pmc.txt 2>&1
tail -40 gpurun_out/${tag}_msda_pmc.txt
# the relation-head kernel of the same passes (algorithmic bytes: per-query tables + gates 5.7 MB, split weights 0.9 MB,
# outputs 8.2 MB at B = 1, N = 200, T = 7, R = 50)
python3 tools/msda_pmc.py gpurun_out/pmc_${tag} --kernel-regex "rel_head_fwd_x6" --name "rel_head_fwd_x6" --alg-bytes 14800000 \
    --out gpurun_out/${tag}_rel_head_pmc.json > gpurun_out/${tag}_rel_head_pmc.txt 2>&1
tail -12 gpurun_out/${tag}_rel_head_pmc.txt
find gpurun_out/pmc_${tag} -name "*.db" -delete
