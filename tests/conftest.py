import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
