#!/usr/bin/env python3
"""Matrix-pipe busy fraction of the split-bf16 kernels from rocprofv3 --pmc passes (tools/pmc_passes.sh <dir> bench mfma):
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), averaged over the kernel's launches.

    python3 tools/mfma_busy.py gpurun_out/pmc_mfma_r04 --out gpurun_out/r04_x6_mfma_pmc.json
    python3 tools/mfma_busy.py --from-text profiles/r03_x6_mfma_pmc.txt --out profiles/r03_x6_mfma_pmc.json

bench.py attaches the newest profiles/r*_x6_mfma_pmc.json to its split-bf16 roofline entries as `mfma_busy`."""
import argparse
import glob
import json
import os
import re
import sqlite3

KERNELS = {"ffn_x6_kernel": "ffn_x6_kernel<true", "proj_x6_kernel": "proj_x6_kernel",
           "gemm_split_bf16_f32": "gemm_split_bf16_f32", "rel_head_fwd_x6": "rel_head_fwd_x6",
           "wgrad_split_bf16_f32": "wgrad_split_bf16_f32", "enc_bwd_x6": "enc_bwd",
           # backbone kernels of late round 6 (the layer-3 convolution, all tails, the stem)
           "conv3x3_x6_ksplit_kernel": "conv3x3_x6_ksplit_kernel<256", "conv_tail_x6_kernel": "conv_tail_x6_kernel",
           "stem_x6_kernel": "stem_x6_kernel",
           # bf16 model (stress workload, tools/profile_r05_stress.sh)
           "rel_head_fwd_bf16p": "rel_head_fwd_bf16p", "ffn_bf16_kernel": "ffn_bf16_kernel",
           "linear_bf16_rows32": "linear_bf16_rows32"}
NEED = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY")


def from_dbs(paths):
    dbs = []
    for p in paths:
        dbs += sorted(glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True)) if os.path.isdir(p) else [p]
    acc = {}
    for db in dbs:
        c = sqlite3.connect(db)
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            ni = cols.index("kernel_name")
            ci = cols.index("counter_name") if "counter_name" in cols else cols.index("name")
            vi = cols.index("value") if "value" in cols else cols.index("counter_value")
            rows = c.execute("select * from counters_collection").fetchall()
        except (sqlite3.Error, ValueError):
            continue
        for r in rows:
            if r[ci] not in NEED:
                continue
            for key, pat in KERNELS.items():
                if pat in str(r[ni]):
                    a = acc.setdefault(key, {}).setdefault(r[ci], [0, 0.0])
                    a[0] += 1
                    a[1] += float(r[vi])
    return {k: {c: (v[1] / v[0]) for c, v in d.items()} | {"launches": max(v[0] for v in d.values())} for k, d in acc.items()}


def from_text(path):
    out, key = {}, None
    for line in open(path):
        m = re.match(r"==== (\S+)", line)
        if m:
            key = m.group(1)
            continue
        m = re.search(r"(SQ_\w+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
        if m and key and m.group(1) in NEED and m.group(1) not in out.setdefault(key, {}):
            out[key][m.group(1)] = float(m.group(3))
            out[key]["launches"] = int(m.group(2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--from-text", default=None)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    raw = from_text(a.from_text) if a.from_text else from_dbs(a.paths)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kernel_source_hash import source_hash
    kernels = {}
    for k, d in raw.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
            kernels[k] = {"mfma_busy": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"]), 4),
                          "counters": {c: v for c, v in d.items() if c != "launches"}, "launches": d.get("launches"),
                          "source_sha256": source_hash(k)}   # bench.py quotes the entry only while this still matches
    out = {"definition": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), per-launch averages",
           "source": a.from_text or " ".join(a.paths), "kernels": kernels}
    json.dump(out, open(a.out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:28s} mfma_busy {v['mfma_busy']:.3f}  ({v['launches']} launches)")


if __name__ == "__main__":
    main()
