#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun; copy the summaries from gpurun_out/ to profiles/):
#   kernel stats + forward breakdown of the default bench command, train step by phase / family / kernel (tools/train_gaps.sh),
#   memory counters of the MSDA forward + relation head + fused encoder tail, matrix-pipe busy of the split-bf16 kernels
#   (tools/mfma_busy.py), memory counters of the encoder's MSDA backward pair (tools/pmc_train.sh).
set -u
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python3 bench.py > gpurun_out/${tag}_bench_rocprofv3.log 2>&1
grep -E '^\{"metric"' gpurun_out/${tag}_bench_rocprofv3.log > gpurun_out/${tag}_bench.json
python3 tools/rocpd_stats.py gpurun_out/prof_bench/bench_results.db --top 60 --split-grid msda_fwd_q64:1000 > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_fb -o fb -- python3 bench.py --no-cpu-baseline --extras 0 --steps 20 > /dev/null 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 14 > gpurun_out/${tag}_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_fb
bash tools/train_gaps.sh ${tag} > /dev/null 2>&1
bash tools/pmc_bench.sh ${tag} > /dev/null 2>&1
bash tools/pmc_passes.sh gpurun_out/pmc_x6_${tag} bench mem -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0 > /dev/null 2>&1
python3 tools/msda_pmc.py gpurun_out/pmc_x6_${tag} --kernel-regex 'ffn_x6_kernel<true>' --name 'ffn_x6_kernel<true>' --alg-bytes 42052608 \
    --out gpurun_out/${tag}_ffn_x6_pmc.json > gpurun_out/${tag}_ffn_x6_pmc.txt 2>&1
find gpurun_out/pmc_x6_${tag} -name "*.db" -delete
bash tools/pmc_passes.sh gpurun_out/pmc_mfma_${tag} bench mfma -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0 > /dev/null 2>&1
python3 tools/mfma_busy.py gpurun_out/pmc_mfma_${tag} --out gpurun_out/${tag}_x6_mfma_pmc.json > gpurun_out/${tag}_x6_mfma_pmc.txt 2>&1
find gpurun_out/pmc_mfma_${tag} -name "*.db" -delete
bash tools/pmc_train.sh ${tag} > /dev/null 2>&1
cut -c1-700 gpurun_out/${tag}_bench.json
head -12 gpurun_out/${tag}_train_gaps.txt | cut -c1-200
cat gpurun_out/${tag}_x6_mfma_pmc.txt
ls gpurun_out | grep "^${tag}_"
