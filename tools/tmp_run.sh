set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-400
EGTR_GEMM_SPLIT_BF16=0 timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
