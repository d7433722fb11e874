#!/usr/bin/env python3
"""Evaluator-input post-processing at the VG shape (N = 200, R = 50): egtr_amd.runtime.triplet_candidates (batched
top-k on the device) vs the reference's path restated in oracle/postprocess.py (D2H of pred_rel + full numpy argsort)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egtr_amd.runtime import triplet_candidates  # noqa: E402
from oracle import postprocess as OP  # noqa: E402  (tool: baseline timing only)


def main():
    dev = "cuda:0"
    B, N, C, R = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 200, 150, 50
    g = torch.Generator().manual_seed(0)
    out = {"logits": torch.randn(B, N, C + 1, generator=g).to(dev), "pred_boxes": torch.rand(B, N, 4, generator=g).to(dev),
           "pred_rel": torch.rand(B, N, N, R, generator=g).to(dev), "pred_connectivity": torch.rand(B, N, N, 1, generator=g).to(dev)}
    sizes = torch.tensor([[600, 1000]] * B)
    for _ in range(3):
        r = triplet_candidates(out, C, sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r = triplet_candidates(out, C, sizes)
        _ = [e["pred_rel_inds"].cpu() for e in r]  # what an evaluator would pull to the host
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(3):
        for b in range(B):
            OP.triplet_candidates(out["logits"][b].cpu(), out["pred_boxes"][b].cpu(), out["pred_rel"][b].cpu(),
                                  out["pred_connectivity"][b].cpu(), C, sizes[b])
    t_ref = (time.perf_counter() - t0) / 3
    print(f"post-processing B={B}: device top-k {t_dev * 1e3:.2f} ms per batch, reference path (D2H + numpy argsort) "
          f"{t_ref * 1e3:.1f} ms per batch")


if __name__ == "__main__":
    main()
