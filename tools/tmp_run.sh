set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | grep -v Warning | tail -70
timeout 900 python bench.py --mode train --steps 10 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
