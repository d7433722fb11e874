#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc pass databases of `python bench.py` (tools/pmc_passes.sh ... mem) into the per-launch
memory figures bench.py attaches to its roofline entry: profiles/r02_msda_pmc.json.

    python tools/msda_pmc.py <dir-with-pass-dbs> --kernel-regex 'msda_fwd_q64_f32<true, false' --name 'msda_fwd_q64_f32<fused prologue>' \
        --alg-bytes 44932608 --out gpurun_out/r02_msda_pmc.json

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KB and counts
128-B requests of 16-B-per-lane reads at 64 B -> doubled; WRITE_SIZE is in KB and taken as is (it equals the output
size here).  TCP_TOTAL_CACHE_ACCESSES counts 64-B L1 accesses.  Every figure is the average over the matching launches
of its own pass (counters from different passes are never combined per launch).
"""
import argparse
import glob
import json
import os
import re
import sqlite3


def collect(paths, rx, min_grid=0):
    out, names = {}, {}
    dbs = []
    for p in paths:
        dbs += sorted(glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True)) if os.path.isdir(p) else [p]
    for db in dbs:
        c = sqlite3.connect(db)
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            rows = c.execute("select * from counters_collection").fetchall()
        except sqlite3.Error:
            continue
        ni = cols.index("kernel_name")
        gi = next((cols.index(x) for x in ("grid_size_x", "grid_x", "grid_size") if x in cols), None)
        if min_grid and gi is None:
            print(f"--min-grid: no grid column in {cols}")
        ci = cols.index("counter_name") if "counter_name" in cols else cols.index("name")
        vi = cols.index("value") if "value" in cols else cols.index("counter_value")
        for r in rows:
            kn = str(r[ni])
            if not rx.search(kn):
                continue
            if min_grid and gi is not None and int(r[gi]) < min_grid:
                continue
            names[kn] = names.get(kn, 0) + 1
            s = out.setdefault(r[ci], [0, 0.0])
            s[0] += 1
            s[1] += float(r[vi])
    return {k: (n, tot / n) for k, (n, tot) in out.items()}, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--kernel-regex", required=True)
    ap.add_argument("--name", required=True, help="the kernel label bench.py prints (roofline.kernel)")
    ap.add_argument("--alg-bytes", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--min-grid", type=int, default=0,
                    help="only launches whose grid (work-items in x) is at least this (encoder-shaped launches of a kernel "
                         "that also runs decoder-shaped)")
    a = ap.parse_args()
    avg, names = collect(a.paths, re.compile(a.kernel_regex), a.min_grid)
    print("matching kernels:")
    for n, k in sorted(names.items(), key=lambda x: -x[1]):
        print(f"  {k:6d} rows  {n[:150]}")
    for k, (n, v) in sorted(avg.items()):
        print(f"  {k:40s} n={n:6d} avg={v:16.1f}")
    g = lambda k: avg[k][1] if k in avg else None  # noqa: E731
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kernel_source_hash import files_for, source_hash
    res = {"kernel": a.name, "launches_averaged": {k: n for k, (n, _) in avg.items()},
           "algorithmic_bytes_per_launch": a.alg_bytes or None,
           # provenance: bench.py quotes this file only while the kernel's sources still hash to this value
           "source_files": files_for(a.name), "source_sha256": source_hash(a.name)}
    if g("FETCH_SIZE") is not None:
        res["fetch_size_kb_raw"] = g("FETCH_SIZE")
        res["fetch_bytes_corrected"] = int(g("FETCH_SIZE") * 1024 * 2)
    if g("WRITE_SIZE") is not None:
        res["write_bytes"] = int(g("WRITE_SIZE") * 1024)
    if "fetch_bytes_corrected" in res and "write_bytes" in res:
        res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        res["l2_hit"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
        res["l1_accesses_64B"] = g("TCP_TOTAL_CACHE_ACCESSES_sum")
        res["l1_gather_bytes"] = int(g("TCP_TOTAL_CACHE_ACCESSES_sum") * 64)
    for k in ("TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"):
        if g(k) is not None:
            res[k] = g(k)
    res["how"] = ("rocprofv3 --kernel-trace --pmc, one pass per counter group (tools/pmc_passes.sh ... mem) over "
                  "`python bench.py --no-cpu-baseline`; averages over the launches whose kernel name matches "
                  f"/{a.kernel_regex}/; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as reported. "
                  "The operands are L2 / Infinity-Cache resident between launches; the counter sits on the fabric side "
                  "of L2 and includes Infinity-Cache hits.")
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
