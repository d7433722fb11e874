#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (ROCm 7.2 default output) into a per-kernel stats table
(the same columns as `rocprofv3 --stats`' kernel_stats.csv): calls, total / average / min / max duration, %.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--top 40] [--csv out.csv]
"""
import argparse
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\[clone .*?\]", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--last-ms", type=float, default=0.0,
                    help="only dispatches that start in the last X ms of the trace (steady state after warm-up)")
    ap.add_argument("--split-grid", default=None,
                    help="NAME:THRESHOLD -- report kernels whose name contains NAME separately for launches with at "
                         "least THRESHOLD workgroups (e.g. msda_fwd_q64:1000 = the encoder-shaped MSDA launches)")
    ap.add_argument("--exact-grid", default=None,
                    help="NAME:WORKGROUPS -- the launches of kernels whose name contains NAME with EXACTLY that many workgroups "
                         "(msda_fwd_q64:3135 = the encoder launch of the bench workload, 12 537 queries / 4 per workgroup; the "
                         "default bench command also launches the kernel at the training batch and at the mixed image shapes)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, start, end from kernels").fetchall()
    if a.split_grid:
        nm, thr = a.split_grid.rsplit(":", 1)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        gx = next((x for x in ("grid_x", "grid_size_x", "grid_size") if x in cols), None)
        wx = next((x for x in ("workgroup_x", "workgroup_size_x", "workgroup_size") if x in cols), None)
        if gx is None:
            print(f"--split-grid: no grid column in {cols}")
        else:
            q = f"select name, start, end, {gx}, {wx if wx else 1} from kernels"
            big = [(e - s_) for n, s_, e, g, w in c.execute(q) if nm in n and (g // max(int(w), 1)) >= int(thr)]
            small = [(e - s_) for n, s_, e, g, w in c.execute(q) if nm in n and (g // max(int(w), 1)) < int(thr)]
            for lab, v in ((f">= {thr} workgroups", big), (f"< {thr} workgroups", small)):
                if v:
                    print(f"{nm} launches with {lab}: n = {len(v)}, average {sum(v) / len(v) / 1e3:.2f} us, "
                          f"min {min(v) / 1e3:.2f} us, max {max(v) / 1e3:.2f} us")
    if a.exact_grid:
        nm, want = a.exact_grid.rsplit(":", 1)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        gx = next((x for x in ("grid_x", "grid_size_x", "grid_size") if x in cols), None)
        wx = next((x for x in ("workgroup_x", "workgroup_size_x", "workgroup_size") if x in cols), None)
        if gx is not None:
            q = f"select name, start, end, {gx}, {wx if wx else 1} from kernels"
            v = [(e - s_) for n, s_, e, g, w in c.execute(q) if nm in n and (g // max(int(w), 1)) == int(want)]
            if v:
                v.sort()
                print(f"{nm} launches with exactly {want} workgroups: n = {len(v)}, average {sum(v) / len(v) / 1e3:.2f} us, "
                      f"median {v[len(v) // 2] / 1e3:.2f} us, min {v[0] / 1e3:.2f} us, max {v[-1] / 1e3:.2f} us")
    if a.last_ms > 0 and rows:
        t_end = max(r[2] for r in rows)
        rows = [r for r in rows if r[1] >= t_end - a.last_ms * 1e6]
        busy = sum(r[2] - r[1] for r in rows)
        print(f"window: last {a.last_ms:.0f} ms, GPU busy {100.0 * busy / (a.last_ms * 1e6):.1f} % "
              f"(sum of kernel durations / window)")
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = e - s
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    total = sum(v[1] for v in agg.values()) or 1
    table = sorted(((n, v[0], v[1], v[1] / v[0], v[2], v[3], 100.0 * v[1] / total) for n, v in agg.items()),
                   key=lambda r: -r[2])
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")]
    out += [(short(n), c_, t, round(avg, 1), mn, mx, round(pc, 3)) for n, c_, t, avg, mn, mx, pc in table]
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            csv.writer(f).writerows(out)
    print(f"{len(rows)} dispatches, {len(agg)} distinct kernels, total kernel time {total / 1e6:.3f} ms")
    for r in out[: a.top + 1]:
        print("%-112s %7s %14s %11s %9s %9s %8s" % r)


if __name__ == "__main__":
    sys.exit(main())
