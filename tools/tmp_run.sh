set -u
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warning\|warn" > gpurun_out/suite2.log
grep -n "Error\|^E \|assert\|passed\|failed" gpurun_out/suite2.log | head -40
