#!/usr/bin/env python3
"""Per-kernel averages of PMC counters from rocprofv3 rocpd databases (one db per --pmc pass).
    python tools/pmc_summary.py <dir-with-*_results.db...> --kernel msda_fwd"""
import argparse
import glob
import os
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--kernel", default="msda")
    a = ap.parse_args()
    dbs = []
    for p in a.paths:
        dbs += sorted(glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True)) if os.path.isdir(p) else [p]
    for db in dbs:
        c = sqlite3.connect(db)
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            rows = c.execute("select * from counters_collection").fetchall()
        except sqlite3.Error as e:
            print(db, "no counters_collection:", e)
            continue
        ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else None, None, None
        for cand in ("counter_name", "name"):
            if cand in cols:
                ci = cols.index(cand)
        for cand in ("value", "counter_value"):
            if cand in cols:
                vi = cols.index(cand)
        if ni is None or ci is None or vi is None:
            print(db, "unexpected schema", cols)
            continue
        agg = {}
        for r in rows:
            if a.kernel not in str(r[ni]):
                continue
            k = (str(r[ni]).split("(")[0][-40:], r[ci])
            s = agg.setdefault(k, [0, 0.0])
            s[0] += 1
            s[1] += float(r[vi])
        print(f"== {db}")
        for (kn, cn), (n, tot) in sorted(agg.items()):
            print(f"  {kn:40s} {cn:40s} n={n:5d} avg={tot / n:16.1f}")


if __name__ == "__main__":
    main()
