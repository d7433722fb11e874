#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun; copy the summaries from gpurun_out/ to profiles/):
#   kernel stats / forward breakdown / train steady state of the default command (tools/profile_bench.sh),
#   memory counters of the MSDA forward + relation head (tools/pmc_bench.sh) and of the fused FFN / projection kernels,
#   matrix-core busy counters of the x6 kernels, memory counters of the encoder's MSDA backward pair (tools/pmc_train.sh).
set -u
tag=${1:-r03}
bash tools/profile_bench.sh $tag
bash tools/pmc_bench.sh $tag
cd "$GRAFT_REPO_ROOT"
# the fused FFN / projection kernels from the same memory passes cannot be re-read (dbs deleted): own short passes
bash tools/pmc_passes.sh gpurun_out/pmc_x6_${tag} bench mem -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0
python3 tools/msda_pmc.py gpurun_out/pmc_x6_${tag} --kernel-regex 'ffn_x6_kernel<true>' --name 'ffn_x6_kernel<true>' --alg-bytes 42052608 \
    --out gpurun_out/${tag}_ffn_x6_pmc.json > gpurun_out/${tag}_ffn_x6_pmc.txt 2>&1
python3 tools/msda_pmc.py gpurun_out/pmc_x6_${tag} --kernel-regex 'proj_x6_kernel' --name 'proj_x6_kernel' \
    --out gpurun_out/${tag}_proj_x6_pmc.json >> gpurun_out/${tag}_ffn_x6_pmc.txt 2>&1
find gpurun_out/pmc_x6_${tag} -name "*.db" -delete
bash tools/pmc_passes.sh gpurun_out/pmc_mfma_${tag} bench mfma -- python3 bench.py --no-cpu-baseline --extras 0 --steps 3 --warmup 2 --tune-gemm 0
for k in ffn_x6_kernel proj_x6_kernel gemm_split_bf16_f32 rel_head_fwd_x6; do
  echo "==== $k" >> gpurun_out/${tag}_x6_mfma_pmc.txt
  python3 tools/pmc_summary.py gpurun_out/pmc_mfma_${tag} --kernel $k >> gpurun_out/${tag}_x6_mfma_pmc.txt 2>&1
done
find gpurun_out/pmc_mfma_${tag} -name "*.db" -delete
bash tools/pmc_train.sh $tag
ls -la gpurun_out | grep ${tag}_ | head -40
