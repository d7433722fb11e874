cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "stem_fused_bf16 or bf16_bottleneck_tail" 2>&1 | tail -2
for i in 1 2; do for f in 0 1; do
  echo "== STEM_FUSED_BF16=$f run $i"
  python3 -c "import sys, runpy; import egtr_amd.backbone as b; b.STEM_FUSED_BF16 = bool($f); sys.argv = ['stress_bench.py', '--iters', '10']; runpy.run_path('tools/stress_bench.py', run_name='__main__')" 2>&1 | grep "HIP graph"
done; done
