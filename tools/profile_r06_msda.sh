#!/bin/bash
# Round 6, after the prologue trim of msda_fwd_q64_f32: the counter profiles that hash csrc/msda.hip again (MSDA forward fp32 +
# relation head of the same passes, MSDA backward pair, bf16 MSDA of the stress workload) and the default bench line on top.
set -u
tag=r06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputest.log 2>&1; tail -2 gpurun_out/${tag}_gputest.log
bash tools/pmc_bench.sh ${tag} > /dev/null 2>&1
bash tools/pmc_train.sh ${tag} > /dev/null 2>&1
bash tools/pmc_passes.sh gpurun_out/pmc_stress_${tag} stress mem -- python3 tools/stress_bench.py --iters 2 > gpurun_out/${tag}_stress_pmc_passes.log 2>&1
python3 tools/msda_pmc.py gpurun_out/pmc_stress_${tag} --kernel-regex 'msda_fwd_q32_bf16<true' --name 'msda_fwd_q32_bf16<fused prologue>' \
    --alg-bytes 637177856 --min-grid 1000000 --out gpurun_out/${tag}_msda_bf16_pmc.json > gpurun_out/${tag}_msda_bf16_pmc.txt 2>&1
find gpurun_out -name "*.db" -delete
bash tools/pmc_passes.sh gpurun_out/pmc_sq_${tag} bench sq -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0 > /dev/null 2>&1
for k in msda_fwd_q64 ffn_x6_kernel gemm_split_bf16 rel_head_fwd_x6 decoder_layer_cluster; do echo "=== $k"; python3 tools/pmc_summary.py gpurun_out/pmc_sq_${tag} --kernel $k 2>&1 | tail -30; done > gpurun_out/${tag}_sq_pmc.txt 2>&1
find gpurun_out -name "*.db" -delete
cp gpurun_out/${tag}_msda_pmc.json gpurun_out/${tag}_rel_head_pmc.json gpurun_out/${tag}_msda_bwd_pmc.json gpurun_out/${tag}_msda_bf16_pmc.json profiles/ 2>/dev/null
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cut -c1-1300 gpurun_out/${tag}_bench.json; grep -A3 "msda_fwd_q64" gpurun_out/${tag}_sq_pmc.txt | head -12
