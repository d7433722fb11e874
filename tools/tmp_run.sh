set -u
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "clamp_nonfinite" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "loss_and_grads or train_step or graphed" 2>&1 | tail -3
timeout 900 python bench.py --mode train --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c90-200
