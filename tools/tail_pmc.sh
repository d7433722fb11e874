#!/bin/bash
# Counters of the backbone kernels of late round 6 (separate --pmc passes, tools/pmc_passes.sh): matrix-pipe busy of the x6 kernels
# over the bench forward (-> ${tag}_x6_mfma_pmc.json, all split-bf16 kernels), and memory counters: the bf16 tail kernel inside the
# stress forward (bs 16, 800x1333; layer 1 = conv_tail_bf16_kernel<4>: 1 068 800 rows x (64 + 256 + 256) x 2 bytes) and the
# fp32 kernel inside the bench forward (layer 1 = conv_tail_x6_kernel<32, 1, 4>: 37 500 rows x (64 + 256 + 256) x 4 bytes).
set -u
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_tail_${tag}
bash tools/pmc_passes.sh gpurun_out/pmc_tail_${tag}/stress stress mem -- python3 tools/stress_bench.py --iters 2 --find 0 > gpurun_out/${tag}_tail_pmc_passes.log 2>&1
bash tools/pmc_passes.sh gpurun_out/pmc_tail_${tag}/bench bench mem -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-probes --extras 0 >> gpurun_out/${tag}_tail_pmc_passes.log 2>&1
{
python3 tools/msda_pmc.py gpurun_out/pmc_tail_${tag}/stress --kernel-regex 'conv_tail_bf16_kernel<4>' --name 'conv_tail_bf16_kernel<4>' \
  --alg-bytes $((1068800 * 576 * 2)) --out gpurun_out/${tag}_conv_tail_bf16_pmc.json
python3 tools/msda_pmc.py gpurun_out/pmc_tail_${tag}/stress --kernel-regex 'conv_tail_bf16_kernel<16>' --name 'conv_tail_bf16_kernel<16>' \
  --alg-bytes $((67200 * 2304 * 2))
python3 tools/msda_pmc.py gpurun_out/pmc_tail_${tag}/bench --kernel-regex 'conv_tail_x6_kernel<32, 1, 4, 4>' --name 'conv_tail_x6_kernel<32, 1, 4, 4>' \
  --alg-bytes $((37500 * 576 * 4)) --out gpurun_out/${tag}_conv_tail_x6_pmc.json
python3 tools/msda_pmc.py gpurun_out/pmc_tail_${tag}/bench --kernel-regex 'conv_tail_x6_kernel<32, 1, 16, 4>' --name 'conv_tail_x6_kernel<32, 1, 16, 4>' \
  --alg-bytes $((2394 * 2304 * 4))
python3 tools/msda_pmc.py gpurun_out/pmc_tail_${tag}/bench --kernel-regex 'conv3x3_x6_ksplit_kernel<256, 128, 1>' --name 'conv3x3_x6_ksplit_kernel<256, 128, 1>' \
  --alg-bytes $((4 * (2 * 38 * 63 * 256 + 9 * 256 * 256))) --out gpurun_out/${tag}_conv3x3_x6_pmc.json
python3 tools/msda_pmc.py gpurun_out/pmc_tail_${tag}/bench --kernel-regex 'stem_x6_kernel' --name 'stem_x6_kernel' \
  --alg-bytes $((4 * (3 * 600 * 1000 + 150 * 250 * 64))) --out gpurun_out/${tag}_stem_x6_pmc.json
} > gpurun_out/${tag}_conv_tail_pmc.txt 2>&1
# (SQ counter passes serialise the kernels: a handful of forwards is all that fits the 150 s limit of a pass)
bash tools/pmc_passes.sh gpurun_out/pmc_tail_${tag}/mfma bench mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-probes --extras 0 >> gpurun_out/${tag}_tail_pmc_passes.log 2>&1
python3 tools/mfma_busy.py gpurun_out/pmc_tail_${tag}/mfma --out gpurun_out/${tag}_x6_mfma_pmc.json > gpurun_out/${tag}_x6_mfma_pmc.txt 2>&1
cat gpurun_out/${tag}_x6_mfma_pmc.txt
rm -rf gpurun_out/pmc_tail_${tag}
grep -E "kernel\"|hbm_bytes|algorithmic|l2_hit|l1_gather" gpurun_out/${tag}_conv_tail_pmc.txt
