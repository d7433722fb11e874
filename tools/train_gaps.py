#!/usr/bin/env python3
"""Where does the bs = 4 train step (bench.py --mode train) spend its wall time, and is the GPU the bottleneck?

Two runs of the SAME loop, merged by a third invocation:

  python3 tools/train_gaps.py measure gpurun_out/train_gaps.json
        untraced: HIP events recorded on the launch stream at the phase boundaries (forward | matcher + loss | backward |
        clip + AdamW), no host synchronisation inside the step.  Per phase: GPU wall (event to event), host enqueue time
        (perf_counter at the same points) and the host's LEAD over the GPU at the boundary (> 0: the GPU still has queued
        work when the host gets there; <= 0: the GPU ran dry and waited for the host).
  rocprofv3 --kernel-trace -d <dir> -o tg -- python3 tools/train_gaps.py trace
        traced: the same loop; a marker kernel (an int16 fill, a dtype nothing else in the step fills) is launched at every
        phase boundary so that the kernel trace can be cut into phases without any other trace domain.
  python3 tools/train_gaps.py report gpurun_out/train_gaps.json <dir>/tg_results.db > profiles/rNN_train_gaps.txt
        per phase: untraced wall, sum of kernel durations (traced run), idle = wall - busy, busy fraction; kernel time by
        family; the largest gaps of the traced run for orientation (tracer-inflated).
"""
import json
import os
import re
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["forward (backbone, encoder, decoder, heads)", "matcher + SGG loss", "backward", "clip + AdamW + zero_grad"]
WARM, STEPS = 8, 10


def _setup():
    import torch
    import bench
    from egtr_amd.runtime import configure_optimizers, enable_gemm_tuning
    enable_gemm_tuning()
    dev = torch.device("cuda", 0)
    model, cfg, _ = bench.build_model(dev, {"dropout": 0.1})
    model.train()
    opt = configure_optimizers(model, lr=2e-6, lr_backbone=2e-7, lr_initialized=None, weight_decay=1e-4)
    torch.manual_seed(100)
    batch = {"pixel_values": torch.randn(4, 3, bench.H_IMG, bench.W_IMG, device=dev),
             "pixel_mask": torch.ones(4, bench.H_IMG, bench.W_IMG, dtype=torch.long, device=dev),
             "labels": bench.make_targets(4, cfg, dev, 7)}
    return torch, model, opt, batch


def _loop(torch, model, opt, batch, boundary, steps):
    """`boundary(i)` is called at the start of phase i (0..3) and with 4 at the end of the step."""
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    orig_loss = model._loss

    def loss_hook(*a, **kw):
        boundary(1)
        return orig_loss(*a, **kw)

    model._loss = loss_hook
    params = [p for p in model.parameters()]
    try:
        for _ in range(steps):
            DeformableDetrHungarianMatcher.raise_if_invalid()
            boundary(0)
            out = model(pixel_values=batch["pixel_values"], pixel_mask=batch["pixel_mask"], labels=batch["labels"],
                        output_attentions=False, output_attention_states=True, output_hidden_states=True)
            boundary(2)
            out.loss.backward()
            boundary(3)
            torch.nn.utils.clip_grad_norm_(params, 0.1)
            opt.step()
            opt.zero_grad(set_to_none=True)
            boundary(4)
    finally:
        model._loss = orig_loss


def measure(path):
    torch, model, opt, batch = _setup()
    _loop(torch, model, opt, batch, lambda i: None, WARM)
    torch.cuda.synchronize()
    ev, host = [], []

    def boundary(i):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append(e)
        host.append(time.perf_counter())

    t_sync = time.perf_counter()
    _loop(torch, model, opt, batch, boundary, STEPS)
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    # GPU time line: event k completed at g[k] (ms after event 0); host time line: host[k].  The two clocks are tied
    # together at the start (the device was idle when event 0 was recorded: it completed at once).
    g = [ev[0].elapsed_time(e) for e in ev]
    h = [(t - host[0]) * 1e3 for t in host]
    per = 5
    rows = []
    for s in range(STEPS):
        b = s * per
        rows.append({"gpu_ms": [g[b + k + 1] - g[b + k] for k in range(4)],
                     "host_ms": [h[b + k + 1] - h[b + k] for k in range(4)],
                     "lead_ms": [g[b + k] - h[b + k] for k in range(5)],
                     "gpu_gap_to_next_step_ms": (g[b + per] - g[b + 4]) if s + 1 < STEPS else None})
    out = {"steps": STEPS, "warmup": WARM, "wall_ms_per_step": (t_end - t_sync) * 1e3 / STEPS,
           "gpu_span_ms_per_step": (g[-1] - g[0]) / STEPS, "rows": rows, "phases": PHASES}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("wall_ms_per_step", "gpu_span_ms_per_step")}))


def trace():
    torch, model, opt, batch = _setup()
    _loop(torch, model, opt, batch, lambda i: None, WARM)
    torch.cuda.synchronize()
    mark = torch.empty(4096, dtype=torch.int16, device="cuda")

    def boundary(i):
        mark.fill_(i)

    _loop(torch, model, opt, batch, boundary, STEPS)
    torch.cuda.synchronize()


def family(n):
    n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n))
    rules = [("msda_bwd", "egtr: MSDA backward"), ("msda_fwd", "egtr: MSDA forward"), ("msda_geom", "egtr: MSDA geometry"),
             ("gemm_split_bf16", "egtr: split-bf16 GEMM (token linears fwd / dgrad)"),
             ("wgrad_", "egtr: split-bf16 weight gradient"), ("tile_weights", "egtr: weight re-tiling"),
             ("ffn_x6|proj_x6|enc_", "egtr: encoder row-panel kernels"),
             ("rel_head|rel_loss|conn_loss|det_loss|hungarian", "egtr: relation head / losses / matcher"),
             ("linear_skinny", "egtr: skinny linears"), ("self_attn", "egtr: decoder self-attention"),
             ("colsum|column_sum|weighted_col", "egtr: column sums (bias gradients, ReLU masks)"),
             ("add_layernorm|layernorm|partial_final", "egtr: LayerNorm fwd / bwd"), ("bias_act|maxpool|groupnorm|level_geom|sine_pos|clamp|nonfinite|box_decode|pad_batch",
                                                                     "egtr: other elementwise"),
             ("miopen|igemm|Sp3AsmConv|batched_transpose|SubTensorOp|gridwise|naive_conv|Im2Col|Col2Im|transpose_", "vendor: MIOpen convolutions + layout"),
             ("Cijk_|rocblas|hipblaslt", "vendor: rocBLAS / hipBLASLt GEMM"),
             ("multi_tensor_apply|FusedOptimizer|lpnorm", "ATen: optimizer / clip (multi-tensor)"),
             ("fused_dropout|masked_scale|bernoulli", "ATen: dropout"), ("at::native", "ATen: elementwise / reduce / copy / index"),
             ("rocclr", "runtime: copy / fill")]
    for pat, name in rules:
        if re.search(pat, n):
            return name
    return "other: " + n[:50]


def report(jpath, db):
    m = json.load(open(jpath))
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "FillFunctor<short>" in r[0]]
    per = 5
    nsteps = len(marks) // per
    if nsteps < 1:
        raise SystemExit("no marker kernels in the trace")
    busy = [0.0] * 4
    wall_tr = [0.0] * 4
    fam = {}
    fam_n = {}
    kern = {}
    gaps = []
    for s in range(nsteps):
        for k in range(4):
            a, b = marks[s * per + k], marks[s * per + k + 1]
            wall_tr[k] += rows[b][1] - rows[a][2]
            cur_end = rows[a][2]
            for n, st, en in rows[a + 1:b]:
                busy[k] += en - st
                f = family(n)
                fam[f] = fam.get(f, 0.0) + (en - st)
                fam_n[f] = fam_n.get(f, 0) + 1
                kk = kern.setdefault(n, [0, 0.0])
                kk[0] += 1
                kk[1] += en - st
                if st > cur_end:
                    gaps.append((st - cur_end, k, n))
                cur_end = max(cur_end, en)
    busy = [b / nsteps / 1e6 for b in busy]
    wall_tr = [w / nsteps / 1e6 for w in wall_tr]
    R = m["rows"]
    n = len(R)
    gpu = [sum(r["gpu_ms"][k] for r in R) / n for k in range(4)]
    host = [sum(r["host_ms"][k] for r in R) / n for k in range(4)]
    lead = [sum(r["lead_ms"][k] for r in R) / n for k in range(5)]
    print(f"bs = 4 train step, 600x1000, dropout 0.1 (bench.py --mode train): {m['steps']} untraced steps after {m['warmup']} warm-up")
    print(f"untraced wall per step {m['wall_ms_per_step']:.2f} ms (host clock around the loop + one synchronize); "
          f"GPU span first-to-last event {m['gpu_span_ms_per_step']:.2f} ms per step")
    print(f"traced run (rocprofv3 --kernel-trace): {nsteps} steps cut at marker kernels; kernel durations are the tracer's")
    print()
    print(f"{'phase':48s} {'wall (untraced, hipEvent)':>26s} {'busy (sum of kernels)':>22s} {'idle':>8s} {'busy/wall':>10s} "
          f"{'host enqueue':>13s} {'host lead at start':>19s} {'traced wall':>12s}")
    for k in range(4):
        idle = gpu[k] - busy[k]
        print(f"{PHASES[k]:48s} {gpu[k]:23.2f} ms {busy[k]:19.2f} ms {idle:5.2f} ms {busy[k] / gpu[k]:10.3f} "
              f"{host[k]:10.2f} ms {lead[k]:16.2f} ms {wall_tr[k]:9.2f} ms")
    tb, tg = sum(busy), sum(gpu)
    print(f"{'step':48s} {tg:23.2f} ms {tb:19.2f} ms {tg - tb:5.2f} ms {tb / tg:10.3f} {sum(host):10.2f} ms "
          f"{lead[4]:16.2f} ms (end) {sum(wall_tr):6.2f} ms")
    print()
    print("host lead = (GPU completion time of the boundary event) - (host time when it recorded the event): while it is positive the GPU "
          "has queued work;\nthe phases whose lead shrinks are the ones where the host enqueues slower than the GPU executes.")
    print()
    print("kernel time by family (traced run, ms per step, launches per step):")
    for f, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"  {v / nsteps / 1e6:7.3f} ms {fam_n[f] / nsteps:7.1f}  {f}")
    print()
    print("kernels of one step (traced run, averages over the steps; name, launches per step, us per launch, ms per step, % of kernel time):")
    tot = sum(v[1] for v in kern.values())
    for nm, (cnt, dur) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"  {re.sub(r'^void ', '', re.sub(r'.anonymous namespace.::', '', nm))[:100]:100s} {cnt / nsteps:7.1f} {dur / cnt / 1e3:9.1f} "
              f"{dur / nsteps / 1e6:8.3f} {100.0 * dur / tot:6.2f}")
    print()
    print("largest GPU-idle gaps of the TRACED run (tracer-inflated; orientation only), us / phase / next kernel:")
    for d, k, nm in sorted(gaps, reverse=True)[:12]:
        print(f"  {d / 1e3:8.1f}  {PHASES[k][:24]:24s} {family(nm)[:40]:40s} {re.sub(r'^void ', '', nm)[:80]}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "measure":
        measure(sys.argv[2])
    elif len(sys.argv) >= 2 and sys.argv[1] == "trace":
        trace()
    elif len(sys.argv) >= 4 and sys.argv[1] == "report":
        report(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
