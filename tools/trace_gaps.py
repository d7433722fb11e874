#!/usr/bin/env python3
"""Largest idle gaps between consecutive kernels in the last N ms of a rocprofv3 rocpd trace (where does the GPU wait for
the host?).  python tools/trace_gaps.py <db> [--last-ms 150] [--top 25]"""
import argparse
import re
import sqlite3


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))
    return n[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last-ms", type=float, default=150.0)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    t_end = max(r[2] for r in rows)
    rows = [r for r in rows if r[1] >= t_end - a.last_ms * 1e6]
    gaps = []
    cur_end = rows[0][2]
    prev = rows[0][0]
    for n, s, e in rows[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n, (s - rows[0][1]) / 1e6))
        if e > cur_end:
            cur_end, prev = e, n
    tot = sum(g[0] for g in gaps)
    print(f"window {a.last_ms:.0f} ms: {len(rows)} kernels, idle {tot / 1e6:.2f} ms in {len(gaps)} gaps; "
          f"gaps > 50 us: {sum(g[0] for g in gaps if g[0] > 5e4) / 1e6:.2f} ms")
    for g in sorted(gaps, reverse=True)[:a.top]:
        print(f"  {g[0] / 1e3:8.1f} us at t = {g[3]:7.2f} ms   after {short(g[1])}   before {short(g[2])}")


if __name__ == "__main__":
    main()
