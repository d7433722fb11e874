#!/bin/bash
# PMC passes (memory counters) over the default bench command, summarised into gpurun_out/<tag>_msda_pmc.json
# (copy to profiles/ after review).  Usage: bash tools/pmc_bench.sh <tag> <kernel-regex> <kernel-label>
set -u
tag=${1:-r02}; rx=${2:-'msda_fwd_q64_f32<true, false'}; label=${3:-'msda_fwd_q64_f32<fused prologue>'}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/pmc_passes.sh gpurun_out/pmc_${tag} bench mem -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --tune-gemm 0
python3 tools/msda_pmc.py gpurun_out/pmc_${tag} --kernel-regex "$rx" --name "$label" --alg-bytes 44932608 \
    --out gpurun_out/${tag}_msda_pmc.json > gpurun_out/${tag}_msda_pmc.txt 2>&1
tail -40 gpurun_out/${tag}_msda_pmc.txt
find gpurun_out/pmc_${tag} -name "*.db" -delete
