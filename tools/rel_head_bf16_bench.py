#!/usr/bin/env python3
"""Relation head of the bf16 stress shape (B = 16, N = 300, T = 9 slots, R = 50): rel_head_fwd_bf16w (fp32 VALU layer 1)
against rel_head_fwd_bf16p (all layers on the matrix cores, packed tables, W2 in LDS).  hipEvent timing of the wrapper,
which for the packed kernel includes the two table-pack launches."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egtr_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--slots", type=int, default=9)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, N, T, R = a.batch, a.queries, a.slots, 50
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
    d = dict(gate_q=r(B, N, T), gate_k=r(B, N, T), uq=r(B, N, T, 512, sc=0.5).bfloat16(), uk=r(B, N, T, 512, sc=0.5).bfloat16(),
             b1=r(512, sc=0.1), w2r=r(256, 256, sc=1 / 16).bfloat16(), b2r=r(256, sc=0.1), w3r=r(R, 256, sc=1 / 16).bfloat16(),
             b3r=r(R, sc=0.1), w2c=r(256, 256, sc=1 / 16).bfloat16(), b2c=r(256, sc=0.1), w3c=r(1, 256, sc=1 / 16).bfloat16(),
             b3c=r(1, sc=0.1))
    outs = {}
    for packed in (False, True):
        ops.REL_HEAD_BF16_PACKED = packed
        for _ in range(2):
            rel, conn, _ = ops.relation_head_bf16w(*d.values())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            rel, conn, _ = ops.relation_head_bf16w(*d.values())
        e1.record()
        torch.cuda.synchronize()
        outs[packed] = (rel, conn)
        print(f"{'rel_head_fwd_bf16p (packed tables)' if packed else 'rel_head_fwd_bf16w (VALU layer 1) '}: "
              f"{e0.elapsed_time(e1) / a.iters * 1e3:8.1f} us per call  (B {B}, N {N}, T {T})")
    dr = (outs[True][0] - outs[False][0]).abs().max().item()
    dc = (outs[True][1] - outs[False][1]).abs().max().item()
    print(f"max |difference| between the two kernels: relation {dr:.3e}, connectivity {dc:.3e} "
          f"(logit scale {outs[False][0].abs().max().item():.2f})")


if __name__ == "__main__":
    main()
