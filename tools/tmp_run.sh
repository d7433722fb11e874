set -u
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py 2>gpurun_out/bench_stderr.log | tee gpurun_out/bench_x6.json | cut -c1-3000
tail -3 gpurun_out/bench_stderr.log
