#!/bin/bash
# Round-5 profile set (gpurun; copy the summaries from gpurun_out/ to profiles/): kernel stats + forward breakdown of the default
# bench command, train step by phase / family / kernel, memory counters of the fp32 MSDA forward + relation head + encoder
# tail and of the new decoder-layer kernel, matrix-pipe busy of the split-bf16 kernels, memory counters of the MSDA backward
# pair, and the stress workload's MSDA (bf16) memory counters with a FETCH_SIZE pass that gets the time it needs.
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python3 bench.py > gpurun_out/${tag}_bench_rocprofv3.log 2>&1
grep -E '^\{"metric"' gpurun_out/${tag}_bench_rocprofv3.log > gpurun_out/${tag}_bench_under_rocprofv3.json
python3 tools/rocpd_stats.py gpurun_out/prof_bench/bench_results.db --top 60 --split-grid msda_fwd_q64:1000 --exact-grid msda_fwd_q64:3135 > gpurun_out/${tag}_bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_bench
bash tools/train_gaps.sh ${tag} > /dev/null 2>&1
bash tools/pmc_bench.sh ${tag} > /dev/null 2>&1
bash tools/pmc_passes.sh gpurun_out/pmc_x6_${tag} bench mem -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0 > /dev/null 2>&1
python3 tools/msda_pmc.py gpurun_out/pmc_x6_${tag} --kernel-regex 'ffn_x6_kernel<true>' --name 'ffn_x6_kernel<true>' --alg-bytes 42052608 \
    --out gpurun_out/${tag}_ffn_x6_pmc.json > gpurun_out/${tag}_ffn_x6_pmc.txt 2>&1
# decoder layer kernel: algorithmic bytes per launch = this layer's weights once (3.8 MB) + q / k / v in and out, states, the
# gathered value lines (200 x 8 x 64 x 128 B = 13.1 MB upper bound)
python3 tools/msda_pmc.py gpurun_out/pmc_x6_${tag} --kernel-regex 'decoder_layer_cluster_f32' --name 'decoder_layer_cluster_f32' --alg-bytes 18000000 \
    --out gpurun_out/${tag}_dec_layer_pmc.json > gpurun_out/${tag}_dec_layer_pmc.txt 2>&1
find gpurun_out/pmc_x6_${tag} -name "*.db" -delete
bash tools/pmc_passes.sh gpurun_out/pmc_mfma_${tag} bench mfma -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0 > /dev/null 2>&1
python3 tools/mfma_busy.py gpurun_out/pmc_mfma_${tag} --out gpurun_out/${tag}_x6_mfma_pmc.json > gpurun_out/${tag}_x6_mfma_pmc.txt 2>&1
find gpurun_out/pmc_mfma_${tag} -name "*.db" -delete
bash tools/pmc_train.sh ${tag} > /dev/null 2>&1
# stress workload, encoder MSDA launch (bf16): the three memory passes, FETCH_SIZE with a 400 s budget
mkdir -p gpurun_out/pmc_stress_${tag}
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/pmc_stress_${tag}/stress_mem$i -o pmc -- python3 tools/stress_bench.py --iters 2 > gpurun_out/pmc_stress_${tag}/stress_mem$i.log 2>&1
  echo "stress pass $i ($ctrs): rc=$?"
done
python3 tools/msda_pmc.py gpurun_out/pmc_stress_${tag} --kernel-regex 'msda_fwd_q32_bf16<true' --name 'msda_fwd_q32_bf16<fused prologue>' \
    --alg-bytes 637177856 --min-grid 1000000 --out gpurun_out/${tag}_msda_bf16_pmc.json > gpurun_out/${tag}_msda_bf16_pmc.txt 2>&1
find gpurun_out/pmc_stress_${tag} -name "*.db" -delete
cut -c1-500 gpurun_out/${tag}_bench_under_rocprofv3.json
head -12 gpurun_out/${tag}_train_gaps.txt | cut -c1-200
cat gpurun_out/${tag}_x6_mfma_pmc.txt | tail -12
tail -12 gpurun_out/${tag}_msda_bf16_pmc.txt
ls gpurun_out | grep "^${tag}_"
