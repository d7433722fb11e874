#!/bin/bash
# Round-5 final measurements (gpurun): the default bench command (untraced) -> gpurun_out/r05_bench.json; matrix-pipe counters of
# the bf16 matrix kernels (tools/stress_mfma_pmc.sh); forward breakdown + MSDA memory counters of the stress workload
# (tools/profile_r05_stress.sh, without its own matrix-pipe pass).  Copy gpurun_out/r05_* to profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
cut -c1-600 gpurun_out/r05_bench.json
tail -3 gpurun_out/r05_bench.err
bash tools/stress_mfma_pmc.sh r05 2>&1 | tail -4
EGTR_SKIP_STRESS_MFMA=1 bash tools/profile_r05_stress.sh r05 2>&1 | tail -25
