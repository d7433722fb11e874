#!/bin/bash
cd "$GRAFT_REPO_ROOT"
{
echo "== correctness with EGTR_GEMM_DB=1"
EGTR_GEMM_DB=1 timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pinning.py tests/test_gpu_train_fused.py -q -k "split_bf16 or gemm or encoder" 2>&1 | tail -3
for rows in 50148 12537; do
  echo "== rows=$rows single buffer"; GEMM_BENCH_ROWS=$rows timeout 300 python tools/gemm_bench.py 2>&1 | grep "^M="
  echo "== rows=$rows ping-pong double buffer"; EGTR_GEMM_DB=1 GEMM_BENCH_ROWS=$rows timeout 300 python tools/gemm_bench.py 2>&1 | grep "^M="
done
} > gpurun_out/r05_gemm_ab.txt 2>&1
cat gpurun_out/r05_gemm_ab.txt
