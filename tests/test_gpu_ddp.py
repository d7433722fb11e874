"""GPU (-m gpu): the RCCL / DistributedDataParallel train step on the GPU box (one GPU: the process group has ONE rank).
The 8-GPU curve is the driver's to measure; this exercises everything a single GPU can: `torch.distributed.run` launch,
nccl (= RCCL) process-group initialisation, an all-reduce, and the DDP reducer over the ctypes-launched autograd
Functions, compared with the unwrapped step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.mark.timeout(900)
def test_ddp_wrapped_step_equals_plain_step_on_one_gpu(tmp_path):
    out = str(tmp_path / "ddp.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a CHILD process: this one has initialised the GPU and must not be replaced or forked into workers
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "_ddp_worker.py"), out],
                       capture_output=True, text=True, timeout=850, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.load(open(out))
    assert d["ranks"] == 1 and d["n_grads"] > 100 and d["grads_missing_under_ddp"] == []
    # same kernels: the wrapper only adds the (one-rank) bucket all-reduce and a division by world size 1; the gradients
    # differ by the order of the float atomics of the decoder-shaped MSDA backward (run to run as well)
    assert d["loss_ddp"] == pytest.approx(d["loss_plain"], rel=1e-6)
    assert d["grad_max_diff"] < 1e-5 * d["grad_scale"] and d["param_max_diff"] < 1e-6, d
    # accumulate = 2: the micro-step before the boundary runs under no_sync (NO collective), the boundary step all-reduces
    # every gradient exactly once, in ceil(gradient bytes / 25 MB bucket cap)-ish buckets (one bucketed pass per optimizer step)
    micro = d["comm"]["allreduces_per_micro_step"]
    assert [stepped for _, stepped in micro] == [False, True, False, True]
    assert micro[0][0] == 0 and micro[2][0] == 0, micro
    gb = d["comm"]["grad_bytes"]
    for n_buckets in (micro[1][0], micro[3][0]):     # (DDP re-cuts its buckets after the first backward: 1, then ~bytes / 25 MB)
        assert 1 <= n_buckets <= gb // (25 * 2 ** 20) + 3, (micro, gb)     # bucketed, not one collective per parameter
    assert 2 * gb <= d["comm"]["allreduce_bytes_total"] <= 2.02 * gb       # each gradient crossed the wire once per step


@pytest.mark.timeout(900)
def test_two_ranks_real_kernels_one_gpu(tmp_path):
    """TWO ranks (two fresh child processes, both on cuda:0, gloo) through DataParallelTrainer with accumulate = 2 on the real
    HIP autograd nodes: the synchronised, clipped gradient equals the single-process gradient of the global batch; the
    micro-step under no_sync issues no collective; the weights agree after the step; a cost matrix refused on ONE rank
    skips the step on BOTH (fused AdamW found_inf) and raises on BOTH at the next step.  (xGMI / RCCL itself stays
    unmeasured on this one-GPU box.)"""
    out = str(tmp_path / "ddp2")
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # CHILD processes: this one has initialised the GPU and must not be replaced or forked into workers
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp2_worker.py"), out],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=850) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
    d0, d1 = (json.load(open(f"{out}.rank{r}.json")) for r in range(2))
    assert d0["n_grads"] > 100 and d0["grads_missing_under_ddp"] == []
    # (fp32: two per-rank sums averaged by the reducer vs four terms added in sequence, float atomics in the MSDA backward and
    # a clip factor computed from each: measured 2e-5 of the largest gradient entry)
    assert d0["grad_max_diff"] < 1e-4 * d0["grad_scale"], d0
    # AdamW's first step moves every weight by ~lr * g / (|g| + eps): where a gradient entry is at rounding level (float atomics
    # in the MSDA backward, MIOpen's convolutions: not bit-reproducible run to run) the two runs may disagree on its SIGN, i.e. by
    # up to two steps (lr = 1e-4).  Measured over repeated runs: 1.5e-5 ... 5.6e-5 -- a bound of half a step was flaky.  The
    # exact statement is the one below: both ranks hold the SAME weights.
    assert d0["param_max_diff"] < 2.1e-4, d0
    for d in (d0, d1):
        micro = d["allreduces_per_micro_step"]
        assert micro[0] == [0, False] and micro[1][0] >= 1 and micro[1][1] is True, micro
        assert d["weights_identical_across_ranks"]
        assert d["weights_unchanged_by_refused_step"]
        assert isinstance(d["raised_at_next_step"], str) and "invalid numeric entries" in d["raised_at_next_step"], d
