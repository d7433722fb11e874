#!/usr/bin/env python3
"""Split ONE graph-replayed forward of bench.py (taken from a rocprofv3 rocpd trace) into stages and list the top
kernels of each.  Stage boundaries: the input projection's GroupNorm (encoder start), the last encoder layer's
closing LayerNorm before the first self-attention launch (decoder start), relation-head launch (end)."""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 12
SEQ = len(sys.argv) > 3  # also print the kernel sequence of each stage
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "rel_head_fwd" in r[0]]
# a bench run also holds eager forwards (warm-up, parity pass) and the stand-alone roofline probes: take the whole
# forwards (>= 80 kernels between two relation-head launches; a replay has ~100 since the backbone kernels of late round 6)
# and of those the one with the shortest span = a graph replay
cands = [rows[idx[k - 1] + 1: idx[k] + 1] for k in range(1, len(idx)) if idx[k] - idx[k - 1] >= 80]
seg = min(cands, key=lambda s_: s_[-1][2] - s_[0][1])
names = [r[0] for r in seg]
print(f"one forward: {len(seg)} kernels, span {(seg[-1][2] - seg[0][1]) / 1e6:.3f} ms, busy {sum(r[2] - r[1] for r in seg) / 1e6:.3f} ms")


def first(pat):
    for i, nm in enumerate(names):
        if pat in nm:
            return i
    return None


def last_before(end, pats, default):
    for i in range(end - 1, -1, -1):
        if any(p in names[i] for p in pats):
            return i + 1
    return default


i_enc = first("msda_fwd")
i_dec = min(i for i in (first("self_attn_fwd"), first("decoder_layer_cluster")) if i is not None)
# the encoder starts after the input projection's GroupNorm + flatten, the decoder after the last encoder layer's closing
# LayerNorm (fused into the FFN kernel or stand-alone)
c_enc = last_before(i_enc, ("gn_apply_flatten",), i_enc - 6)
c_dec = last_before(i_dec, ("ffn_x6_kernel", "add_layernorm"), i_dec - 6)
cuts = [("backbone+input_proj", 0, c_enc), ("encoder", c_enc, c_dec), ("decoder+heads", c_dec, len(seg))]
for label, a, b in cuts:
    s = seg[a:b]
    print(f"\n{label}: {len(s)} kernels, span {(s[-1][2] - s[0][1]) / 1e6:.3f} ms, busy {sum(r[2] - r[1] for r in s) / 1e6:.3f} ms")
    d = defaultdict(lambda: [0, 0])
    for r in s:
        n = re.sub(r"\[clone.*", "", r[0])
        n = re.sub(r"^void ", "", n)[:100]
        d[n][0] += 1
        d[n][1] += r[2] - r[1]
    for n, (cnt, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:TOP]:
        print(f"   {t / 1e3:8.1f} us {cnt:4d}x  {n}")
    if SEQ:
        for r in s:
            print(f"      {(r[2] - r[1]) / 1e3:7.1f}  {re.sub(r'^void ', '', r[0])[:120]}")
