#!/usr/bin/env python3
"""Wall-clock split of one train step (BASELINE configs[2] shape, one GPU): forward (model), loss (matcher + SGG loss),
backward, clip + AdamW -- with a device synchronisation after every phase, so each number is max(CPU, GPU) time of that
phase.  python tools/train_phases.py [--batch 4] [--steps 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--graph", type=int, default=0, help="1 = forward / backward of the static part replayed from HIP graphs")
    a = ap.parse_args()
    from egtr_amd.runtime import configure_optimizers
    dev = torch.device("cuda:0")
    model, cfg, _ = bench.build_model(dev, {"dropout": 0.1})
    model.train()
    opt = configure_optimizers(model, lr=2e-6, lr_backbone=2e-7, lr_initialized=None, weight_decay=1e-4)
    torch.manual_seed(100)
    b = {"pixel_values": torch.randn(a.batch, 3, bench.H_IMG, bench.W_IMG, device=dev),
         "pixel_mask": torch.ones(a.batch, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev),
         "labels": bench.make_targets(a.batch, cfg, dev, 7)}
    names = ["forward without labels", "forward + matcher + loss", "backward", "clip+step"]
    acc = [0.0] * 4

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    kw = dict(pixel_values=b["pixel_values"], pixel_mask=b["pixel_mask"], output_attentions=False,
              output_attention_states=True, output_hidden_states=True)
    if a.graph:
        from egtr_amd.runtime import DataParallelTrainer
        tr = DataParallelTrainer(model, optimizer=opt, accumulate=1, clip=0.1, graph=True)
        acc = [0.0] * 4
        for it in range(a.steps + 3):
            t0 = sync()
            tens = tr._graphed_body(b["pixel_values"], b["pixel_mask"])
            t1 = sync()
            loss, _ = model.loss_from_tensors(tens, b["labels"])
            t2 = sync()
            loss.backward()
            t3 = sync()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            opt.step()
            opt.zero_grad(set_to_none=True)
            t4 = sync()
            if it >= 3:
                for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                    acc[k] += d
        print("graphed train step phases (ms, batch %d): " % a.batch +
              ", ".join(f"{n} {1e3 * v / a.steps:.1f}" for n, v in
                        zip(["forward graph replay", "matcher + loss", "backward (loss eager + graph replay)",
                             "clip+step"], acc)) + f"; step = {1e3 * sum(acc) / a.steps:.1f}")
        return
    for it in range(a.steps + 3):
        t0 = sync()
        out = model(labels=None, **kw)
        del out
        t1 = sync()
        out = model(labels=b["labels"], **kw)
        t2 = sync()
        out.loss.backward()
        t3 = sync()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        opt.zero_grad(set_to_none=True)
        t4 = sync()
        if it >= 3:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[k] += d
    print("train step phases (ms, batch %d): " % a.batch +
          ", ".join(f"{n} {1e3 * v / a.steps:.1f}" for n, v in zip(names, acc)) +
          f"; step = {1e3 * sum(acc[1:]) / a.steps:.1f}")


if __name__ == "__main__":
    main()
