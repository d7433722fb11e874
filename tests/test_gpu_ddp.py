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
