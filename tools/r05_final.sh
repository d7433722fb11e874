#!/bin/bash
# The default bench command (untraced) -> gpurun_out/r05_bench.json, then the stress workload's MSDA (bf16) memory counters.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
cut -c1-900 gpurun_out/r05_bench.json
tail -5 gpurun_out/r05_bench.err
mkdir -p gpurun_out/pmc_stress_r05b
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/pmc_stress_r05b/stress_mem$i -o pmc -- python3 tools/stress_bench.py --iters 2 > gpurun_out/pmc_stress_r05b/stress_mem$i.log 2>&1
  echo "stress pass $i ($ctrs): rc=$?"
done
python3 tools/msda_pmc.py gpurun_out/pmc_stress_r05b --kernel-regex 'msda_fwd_q32_bf16<true' --name 'msda_fwd_q32_bf16<fused prologue>' \
    --alg-bytes 637177856 --min-grid 1000000 --out gpurun_out/r05_msda_bf16_pmc.json > gpurun_out/r05_msda_bf16_pmc.txt 2>&1
find gpurun_out/pmc_stress_r05b -name "*.db" -delete
tail -14 gpurun_out/r05_msda_bf16_pmc.txt
