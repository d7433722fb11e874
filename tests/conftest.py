import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's torch-CPU passes: one thread per usable CPU (a box that shows 100+ cores behind a small CPU quota
    # otherwise runs them 3-4x slower through oversubscription), at most 16
    import torch
    torch.set_num_threads(max(1, min(usable_cpus(), 16)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def cpu_kernels(monkeypatch):
    """Swap the three HIP ops for oracle-built CPU stand-ins (tests/cpu_kernels.py) -- host-logic tests only."""
    import cpu_kernels as ck
    import egtr_amd.ops as ops
    monkeypatch.setattr(ops, "_msda", lambda: ck.OracleMSDA)
    monkeypatch.setattr(ops, "decoder_self_attention", ck.decoder_self_attention)
    monkeypatch.setattr(ops, "relation_head", ck.relation_head)
    return ck
