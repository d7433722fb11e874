#!/bin/bash
# The default bench command (untraced) -> gpurun_out/r05_bench.json, then the matrix-pipe counters of the stress forward's bf16
# matrix kernels (one rocprofv3 --pmc pass over tools/stress_bench.py).  The stress MSDA memory counters and the forward breakdown
# come from tools/profile_r05_stress.sh.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
cut -c1-900 gpurun_out/r05_bench.json
tail -5 gpurun_out/r05_bench.err
mkdir -p gpurun_out/pmc_stress_mfma_r05
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY \
    -d gpurun_out/pmc_stress_mfma_r05/mfma1 -o pmc -- python3 tools/stress_bench.py --iters 1 > gpurun_out/pmc_stress_mfma_r05/mfma1.log 2>&1
echo "stress mfma pass: rc=$?"
python3 tools/mfma_busy.py gpurun_out/pmc_stress_mfma_r05 --out gpurun_out/r05_stress_mfma_pmc.json > gpurun_out/r05_stress_mfma_pmc.txt 2>&1
find gpurun_out/pmc_stress_mfma_r05 -name "*.db" -delete
tail -8 gpurun_out/r05_stress_mfma_pmc.txt
