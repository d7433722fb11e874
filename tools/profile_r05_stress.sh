#!/bin/bash
# Round-5 evidence for the stress workload (BASELINE configs[4] shape on one GPU): forward breakdown of one graph-replayed
# bf16 forward (800x1333, N = 300, 8 decoder layers, bs 16) and the memory counters of its encoder MSDA launch.
# Run through gpurun; copy gpurun_out/r05_stress_* and gpurun_out/r05_msda_bf16_pmc.* to profiles/.
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_stress -o st -- python3 tools/stress_bench.py --iters 4 > gpurun_out/${tag}_stress_run.log 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_stress/st_results.db 24 > gpurun_out/${tag}_stress_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_stress
bash tools/pmc_passes.sh gpurun_out/pmc_stress_${tag} stress mem -- python3 tools/stress_bench.py --iters 2 > gpurun_out/${tag}_stress_pmc_passes.log 2>&1
# algorithmic bytes of one encoder launch at B = 16: S = Lq = 22223, bf16 value + out, bf16 raw offsets / logits (fused entry)
python3 tools/msda_pmc.py gpurun_out/pmc_stress_${tag} --kernel-regex 'msda_fwd_q32_bf16<true' --name 'msda_fwd_q32_bf16<fused prologue>' \
    --alg-bytes 637177856 --min-grid 1000000 --out gpurun_out/${tag}_msda_bf16_pmc.json > gpurun_out/${tag}_msda_bf16_pmc.txt 2>&1
find gpurun_out/pmc_stress_${tag} -name "*.db" -delete
# (matrix-pipe busy of the bf16 matrix kernels: tools/stress_mfma_pmc.sh, from their micro-benchmarks -- a --pmc pass over the whole
# stress forward spends 450 s writing its database and returned no rows for these kernels)
cat gpurun_out/${tag}_stress_forward_breakdown.txt | cut -c1-180
tail -30 gpurun_out/${tag}_msda_bf16_pmc.txt
tail -5 gpurun_out/${tag}_stress_run.log
