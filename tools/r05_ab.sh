#!/bin/bash
# Round-5 A/B batch (gpurun): double-buffered split GEMM, the decoder's two hand-over modes, the bf16 MSDA forward with the
# bit mask.  Output: gpurun_out/r05_ab.txt
cd "$GRAFT_REPO_ROOT"
{
echo "== split GEMM correctness with the double-buffered kernel (EGTR_GEMM_DB=1)"
EGTR_GEMM_DB=1 timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pinning.py -q -k "split_bf16 or gemm" 2>&1 | tail -3
for rows in 50148 12537; do
  echo "== gemm_bench rows=$rows, single buffer"
  GEMM_BENCH_ROWS=$rows timeout 300 python tools/gemm_bench.py 2>&1 | grep "^M="
  echo "== gemm_bench rows=$rows, double buffer"
  EGTR_GEMM_DB=1 GEMM_BENCH_ROWS=$rows timeout 300 python tools/gemm_bench.py 2>&1 | grep "^M="
done
echo "== decoder cluster tests"
timeout 300 python -m pytest tests/test_gpu_decoder_cluster.py -q 2>&1 | tail -3
echo "== decoder layer launch, barrier mode / tagged no-ack mode"
timeout 120 python tools/dec_repro.py 200 6 time 2>&1 | tail -2
EGTR_DECODER_DATAFLOW=1 timeout 120 python tools/dec_repro.py 200 6 time 2>&1 | tail -2
echo "== bf16 tests"
timeout 300 python -m pytest tests -m gpu -q -k "bf16 or stress" 2>&1 | tail -3
echo "== stress bench"
timeout 300 python tools/stress_bench.py --iters 10 2>&1 | grep "stress shape"
} > gpurun_out/r05_ab.txt 2>&1
cat gpurun_out/r05_ab.txt
