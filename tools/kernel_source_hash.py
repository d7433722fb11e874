"""Provenance of counter profiles: the sha256 of the source files a kernel is built from.

bench.py quotes HBM traffic / L2 hit rate / matrix-pipe busy from profiles/r*_pmc.json, which are collected in SEPARATE
rocprofv3 --pmc runs (never inside the timed run).  A profile is only quoted when it was taken from the kernel source that
is in the tree NOW: the summarisers (tools/msda_pmc.py, tools/mfma_busy.py, tools/pmc_train.sh) record this hash next to the
counters, and bench.newest_pmc() / newest_mfma_busy() refuse a file whose hash differs or is missing (a stale
profile whose kernel label still matched was reported as current; file times do not survive a snapshot copy, a hash does)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "egtr_amd", "csrc")

# kernel label prefix (bench.py's roofline.kernel) -> the sources that define the kernel's code
SOURCES = [
    ("msda_bwd", ["msda.hip", "msda_tile.hip", "msda_common.h"]),
    ("msda_fwd", ["msda.hip", "msda_common.h"]),
    ("rel_head_fwd_bf16", ["rel_head_bf16.hip"]),
    ("rel_head", ["rel_head.hip", "xs_format.h", "x6_common.h"]),
    ("conv3x3_x6", ["conv3x3_x6.hip", "xs_format.h", "x6_common.h"]),
    ("stem_x6", ["stem_x6.hip", "xs_format.h", "x6_common.h"]),
    ("conv_tail_bf16", ["conv_tail_bf16.hip", "x6_common.h"]),
    ("conv_tail_x6", ["conv_tail_x6.hip", "xs_format.h", "x6_common.h"]),
    ("ffn_bf16", ["ffn_bf16.hip"]),
    ("ffn_x6", ["ffn_x6.hip", "xs_format.h", "x6_common.h"]),
    ("gemm_split", ["gemm_split.hip", "xs_format.h", "x6_common.h"]),
    ("wgrad_split", ["gemm_split.hip", "xs_format.h", "x6_common.h"]),
    ("decoder_layer_cluster", ["dec_layer.hip", "msda_common.h"]),
]


def files_for(kernel_label):
    for prefix, files in SOURCES:
        if kernel_label.startswith(prefix):
            return files
    return None


def source_hash(kernel_label):
    """sha256 over the kernel's source files (None for a label this table does not know)."""
    files = files_for(kernel_label)
    if files is None:
        return None
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    import sys
    for label in sys.argv[1:]:
        print(label, source_hash(label))
