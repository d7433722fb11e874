"""CPU: libegtr_hip.so builds for gfx950, loads, and exports every symbol include/egtr_hip.h declares.
No compute calls (there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    path = os.path.join(ROOT, "egtr_amd", "libegtr_hip.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "egtr_amd", "csrc"), "-j", "4"], check=True)
    return path


def _declared(headers=("egtr_hip.h", "egtr_hip_test.h")):
    """Every entry point declared under include/ (the drop-in boundary + the test-only variant selectors)."""
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(egtr_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_variant_selectors_are_not_part_of_the_public_boundary():
    public = _declared(("egtr_hip.h",))
    assert not [n for n in public if n.endswith("_variant")]
    assert [n for n in _declared(("egtr_hip_test.h",)) if n.endswith("_variant")]


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for n in ("egtr_msda_forward_f32", "egtr_msda_backward_f32", "egtr_msda_forward_bf16",
              "egtr_self_attn_forward_f32", "egtr_self_attn_backward_f32", "egtr_rel_head_forward_save_f32",
              "egtr_abi_version", "egtr_status_string", "egtr_last_hip_error"):
        assert n in names


def test_library_exports_every_declared_symbol(lib_path):
    h = ctypes.CDLL(lib_path)
    for n in _declared():
        assert hasattr(h, n), f"{n} declared in include/egtr_hip.h but not exported"
    h.egtr_abi_version.restype = ctypes.c_int
    assert h.egtr_abi_version() == 5
    h.egtr_status_string.restype = ctypes.c_char_p
    assert b"ok" == h.egtr_status_string(0)


def test_ctypes_binding_covers_the_header(lib_path):
    """The product binding (egtr_amd/_lib.py) is exactly the public boundary; the test-only selectors of egtr_hip_test.h
    are bound by tests/hip_test_abi.py alone."""
    from egtr_amd import _lib
    import hip_test_abi
    assert sorted(_lib.SIGNATURES) == _declared(("egtr_hip.h",))
    assert sorted(hip_test_abi.SIGNATURES) == _declared(("egtr_hip_test.h",))
    assert not set(_lib.SIGNATURES) & set(hip_test_abi.SIGNATURES)
    _lib.lib()  # resolves every symbol with its argtypes
    hip_test_abi._handle()


def test_null_arguments_are_rejected_without_a_gpu(lib_path):
    """Argument validation happens before any HIP call, so it is checkable on a CPU-only box."""
    from egtr_amd import _lib
    h = _lib.lib()
    st = h.egtr_msda_forward_f32(None, None, None, None, None, None, 1, 1, 8, 32, 4, 1, 4, None)
    assert st == -1
    st = h.egtr_self_attn_forward_f32(None, None, None, None, 1, 1, 8, 32, None, None, None, None)
    assert st == -1
    assert b"invalid" in h.egtr_status_string(-1)
    with pytest.raises(_lib.EgtrHipError):
        _lib.check(st, "x")


def test_usable_cpus_respects_affinity_and_is_applied_to_torch():
    """tests/conftest.py sizes torch's intra-op pool to the CPUs this process may use (the GPU boxes show 256 cores behind
    a 16-CPU quota; the default 128 threads made the oracle passes several times slower)."""
    import os

    import torch

    import conftest
    n = conftest.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert n <= len(os.sched_getaffinity(0))
    assert torch.get_num_threads() == max(1, min(n, 16))
