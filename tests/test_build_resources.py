"""Compile-time resource checks of the LDS-DMA pipeline kernels (no GPU needed: hipcc cross-compiles for gfx950).

ffn_x6_kernel issues the chunk's layer-1 bias as SCALAR loads three stages ahead of their use and waits for them with its own
s_waitcnt -- the compiler does not know the destination SGPRs are still in flight.  That is only sound while it does not spill
them in between, i.e. while the kernel has no SGPR spills at all; a scratch spill of vector registers would likewise put
compiler-generated vector memory operations into the counted vmcnt queue of the DMA ring.  Both are build properties: checked
here so that an edit that breaks them fails on CPU, not as wrong numbers on the GPU."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "egtr_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resources(src, tmp_path):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function",
                          "-Rpass-analysis=kernel-resource-usage", "-c", str(CSRC / src), "-o", str(tmp_path / "o.o")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        m = re.search(r"remark: \S+\s+(\w[\w ]*?)(?: \[bytes/lane\])?: (\d+)", line)
        if m and name:
            res[name][m.group(1).strip()] = int(m.group(2))
    return res


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
@pytest.mark.parametrize("src,kernels,no_sgpr_spill", [
    ("ffn_x6.hip", ("ffn_x6_kernel", "proj_x6_kernel"), ("ffn_x6_kernel",)),
])
def test_pipeline_kernels_have_no_scratch_and_safe_scalar_loads(src, kernels, no_sgpr_spill, tmp_path):
    res = _resources(src, tmp_path)
    for k in kernels:
        hits = {n: r for n, r in res.items() if k in n}
        assert hits, (k, list(res))
        for n, r in hits.items():
            assert r.get("ScratchSize", 0) == 0 and r.get("VGPRs Spill", 0) == 0, (n, r)
            if k in no_sgpr_spill:
                assert r.get("SGPRs Spill", 0) == 0, (n, r)
