"""The HIP kernels keep many loads in flight with inline-asm `global_load` + hand-counted `s_waitcnt vmcnt(N)`.  The
compiler does not know that such a destination register is not valid yet, so a register copy / reuse it inserts between
the load and the wait would silently corrupt data.  tools/check_async_loads.py checks the generated gfx950 assembly:
no instruction may touch the destination of a vector memory load before some vmcnt wait on every path from the load."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_async_loads as lint  # noqa: E402

CSRC = os.path.join(ROOT, "egtr_amd", "csrc")

GOOD = """
kern_a:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tglobal_load_dwordx4 v[2:5], v[0:1], off
\tv_add_f32_e32 v9, v8, v7
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v6, v2, v3
\ts_endpgm
"""
BAD_READ = GOOD.replace("\ts_waitcnt vmcnt(0)\n", "\ts_waitcnt lgkmcnt(0)\n")
BAD_OVERWRITE = """
kern_b:
\tglobal_load_dwordx4 v[2:5], v[0:1], off
\tv_mov_b32_e32 v4, v10
\ts_waitcnt vmcnt(0)
\ts_endpgm
"""
# the wait sits in the loop body that precedes the use on every path, although the use comes first in text order
LOOP_OK = """
kern_c:
\tglobal_load_dwordx4 v[2:5], v[0:1], off
\ts_branch .LBB0_2
.LBB0_1:
\tv_add_f32_e32 v6, v2, v3
\ts_cbranch_vccnz .LBB0_3
.LBB0_2:
\ts_waitcnt vmcnt(0)
\ts_branch .LBB0_1
.LBB0_3:
\ts_endpgm
"""
# a load issued at the end of the loop body is still in flight when the back edge reaches the use
LOOP_BAD = """
kern_d:
\tglobal_load_dwordx4 v[2:5], v[0:1], off
\ts_waitcnt vmcnt(0)
.LBB0_1:
\tv_add_f32_e32 v6, v2, v3
\tglobal_load_dwordx4 v[2:5], v[0:1], off
\ts_cbranch_vccnz .LBB0_1
\ts_waitcnt vmcnt(0)
\ts_endpgm
"""


def test_lint_recognises_premature_use_and_control_flow():
    assert lint.check_asm(GOOD) == []
    assert len(lint.check_asm(BAD_READ)) == 1 and "v2" in lint.check_asm(BAD_READ)[0][3]
    assert len(lint.check_asm(BAD_OVERWRITE)) == 1
    assert lint.check_asm(LOOP_OK) == []
    assert len(lint.check_asm(LOOP_BAD)) >= 1


# LDS-DMA (no destination register) and a load overwriting the dead destination of an earlier load are not findings
DMA_AND_WAW = """
kern_e:
\tglobal_load_lds_dwordx4 v[26:27], off
\tv_add_u32_e32 v27, 0x400, v27
\tglobal_load_dword v16, v[4:5], off
\tglobal_load_dword v16, v[4:5], off
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v6, v16, v3
\ts_endpgm
"""


def test_lint_ignores_lds_dma_and_load_after_load():
    assert lint.check_asm(DMA_AND_WAW) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("src", sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")))
def test_inline_asm_loads_are_waited_for_before_any_use(src):
    """Every kernel source: rel_head.hip (fp32 + bf16 kernels) and linear.hip issue loads through inline asm with
    hand-counted waits; the others only have compiler-managed loads and must pass trivially."""
    findings = lint.check_asm(lint.compile_to_asm(os.path.join(ROOT, "egtr_amd", "csrc", src)))
    assert findings == [], findings[:5]
