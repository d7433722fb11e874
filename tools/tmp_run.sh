set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split" 2>&1 | tail -5
python tools/relhead_bench.py 2>&1 | grep "fwd B"
python tools/relhead_bench.py 4 2>&1 | grep "x6"
