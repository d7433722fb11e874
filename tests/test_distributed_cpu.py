"""CPU, world_size = 2, gloo: the data-parallel train step (egtr_amd/runtime.py::DataParallelTrainer).

The three HIP ops are swapped for the oracle-built stand-ins in every rank (tests/cpu_kernels.py) -- this file
checks the DISTRIBUTED logic only: DDP gradient all-reduce == the single-process gradient of the same global batch
averaged over ranks, no_sync on accumulation micro-steps, identical replicas after the optimizer step, and the
inference sharding (each rank its own images, no collective)."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CFG = dict(num_queries=12, encoder_layers=1, decoder_layers=2, dropout=0.0, auxiliary_loss=False, num_labels=6,
           num_rel_labels=5, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0, connectivity_loss_coefficient=30.0,
           smoothing=1e-14, rel_sample_negatives=80, rel_sample_nonmatching=80, rel_sample_negatives_largest=True,
           rel_sample_nonmatching_largest=True, use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12,
           logit_adjustment=False, logit_adj_tau=0.3)


def _setup_paths():
    for p in (ROOT, HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build():
    _setup_paths()
    import cpu_kernels as ck
    import egtr_amd.ops as ops
    ops._msda = lambda: ck.OracleMSDA
    ops.decoder_self_attention = ck.decoder_self_attention
    ops.relation_head = ck.relation_head
    import egtr_amd.deformable_detr as pdd
    import _ref_import
    import helpers as Hh
    import weights as W
    from egtr_amd.egtr import DetrForSceneGraphGeneration
    cfg = Hh.product_config(CFG)
    orig = pdd.DeformableDetrTimmConvEncoder
    pdd.DeformableDetrTimmConvEncoder = _ref_import.make_stub_backbone_class()
    try:
        torch.manual_seed(0)
        model = DetrForSceneGraphGeneration(cfg, fg_matrix=W.fg_matrix(6, 5))
    finally:
        pdd.DeformableDetrTimmConvEncoder = orig
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = W.fill_state_dict(shapes, seed=5)
    sd["triplet_dist"], sd["rel_dist"] = model.triplet_dist.data.clone(), model.rel_dist.data.clone()
    model.load_state_dict(sd)
    return model.train()


def _batch(seed):
    _setup_paths()
    import weights as W
    rng = W.rng_inputs(seed)
    pv = torch.from_numpy(rng.standard_normal((1, 3, 64, 96))).float()
    pm = torch.ones(1, 64, 96, dtype=torch.long)
    return {"pixel_values": pv, "pixel_mask": pm, "labels": W.make_targets(seed, 1, 12, 6, 5, tmin=2, tmax=4)}


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from egtr_amd.runtime import DataParallelTrainer, init_distributed
    model = _build()
    assert init_distributed() == world and dist.get_backend() == "gloo"
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    tr = DataParallelTrainer(model, optimizer=opt, accumulate=2, clip=1e9)
    # micro-step 1 (no_sync): gradients must stay LOCAL; micro-step 2: all-reduced average of both micro-steps
    grads = {}
    loss1, _, stepped = tr.training_step(_batch(100 + rank))
    assert not stepped
    g_local = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gathered = [None] * world
    dist.all_gather_object(gathered, float(g_local["rel_predictor_gate.weight"].abs().sum()))
    assert abs(gathered[0] - gathered[1]) > 1e-9, "no_sync micro-step must not all-reduce"
    # hook the optimizer to capture the final (synced) gradient before it is consumed
    captured = {}
    orig_step = opt.step

    def step(*a, **k):
        captured.update({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        return orig_step(*a, **k)

    opt.step = step
    loss2, ld, stepped = tr.training_step(_batch(200 + rank))
    assert stepped
    psum = float(sum(p.detach().double().sum() for p in model.parameters()))
    res = dict(rank=rank, loss1=float(loss1), loss2=float(loss2), psum=psum,
               gnorm=float(torch.sqrt(sum((g.double() ** 2).sum() for g in captured.values()))),
               g_gate=captured["rel_predictor_gate.weight"].flatten()[:8].tolist(),
               keys=sorted(ld.keys()))
    # inference sharding: each rank runs its own image, no collective involved
    model.eval()
    with torch.no_grad():
        o = model(pixel_values=_batch(300 + rank)["pixel_values"], pixel_mask=_batch(300 + rank)["pixel_mask"],
                  output_attention_states=True)
    res["pred_rel_sum"] = float(o.pred_rel.double().sum())
    # a cost matrix refused by the device matcher on ONE rank (simulated: the CPU matcher is scipy and raises by itself) must
    # reach every rank: the 4-byte flag is all-reduced (MAX) so that the replicas skip / stop together
    from egtr_amd.deformable_detr import _STEP_MATCHER_STATUS
    _STEP_MATCHER_STATUS.append(torch.tensor([0, 1 if rank == 1 else 0], dtype=torch.int32))
    res["refused_flag"] = float(tr._refused_flag())
    _STEP_MATCHER_STATUS.append(torch.zeros(2, dtype=torch.int32))
    res["clean_flag"] = float(tr._refused_flag())
    res["no_status_flag"] = tr._refused_flag() is None
    # ... and with an optimizer that has no device-side skip (SGD here) the step is stopped on EVERY rank before the weights move
    model.train()
    before = float(sum(p.detach().double().sum() for p in model.parameters()))
    tr._micro = 1                                  # the next micro-step is the accumulation boundary
    orig_common = tr.common_step

    def common_with_refusal(batch):
        out_ = orig_common(batch)
        _STEP_MATCHER_STATUS.append(torch.tensor([2 if rank == 1 else 0], dtype=torch.int32))
        return out_

    tr.common_step = common_with_refusal
    try:
        tr.training_step(_batch(400 + rank))
        res["raised"] = False
    except ValueError:
        res["raised"] = True
    res["psum_unchanged"] = float(sum(p.detach().double().sum() for p in model.parameters())) == before
    res["grads_cleared"] = all(p.grad is None for p in model.parameters())
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = [json.load(open(f"{out}.{i}")) for i in range(world)]
    # replicas stay identical after the synchronised step
    assert abs(r[0]["psum"] - r[1]["psum"]) < 1e-6 * max(1.0, abs(r[0]["psum"]))
    assert abs(r[0]["gnorm"] - r[1]["gnorm"]) < 1e-6 * r[0]["gnorm"] and r[0]["g_gate"] == r[1]["g_gate"]
    assert "rel_gate_0" in r[0]["keys"] and "loss_rel" in r[0]["keys"]
    for k in range(world):   # the refusal of rank 1 is seen by both ranks; the step stops on both, weights intact
        assert r[k]["refused_flag"] == 1.0 and r[k]["clean_flag"] == 0.0 and r[k]["no_status_flag"], r[k]
        assert r[k]["raised"] and r[k]["psum_unchanged"] and r[k]["grads_cleared"], r[k]
    # single-process reference: same 4 micro-batches, gradient = mean over ranks of (sum over micro-steps / accumulate)
    model = _build()
    total = None
    for rank in range(world):
        for seed in (100 + rank, 200 + rank):
            b = _batch(seed)
            model.zero_grad()
            out_ = model(pixel_values=b["pixel_values"], pixel_mask=b["pixel_mask"], labels=b["labels"],
                         output_attention_states=True)
            (out_.loss / 2).backward()
            g = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
            total = g if total is None else {n: total[n] + g[n] for n in g}
    ref = {n: v / world for n, v in total.items()}
    gnorm = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref.values())))
    assert abs(gnorm - r[0]["gnorm"]) < 1e-4 * gnorm, (gnorm, r[0]["gnorm"])
    assert np.allclose(ref["rel_predictor_gate.weight"].flatten()[:8].numpy(), np.array(r[0]["g_gate"]), rtol=1e-4,
                       atol=1e-7)
    # inference shards are independent: different images -> different outputs, each equal to a local run
    assert abs(r[0]["pred_rel_sum"] - r[1]["pred_rel_sum"]) > 1e-6
    model.eval()
    for rank in range(world):
        b = _batch(300 + rank)
        with torch.no_grad():
            o = model.__class__.forward(model, pixel_values=b["pixel_values"], pixel_mask=b["pixel_mask"],
                                        output_attention_states=True)
        # the local model has not taken the SGD step, so only check shape / finiteness here
        assert torch.isfinite(o.pred_rel).all()


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_flag_starts_that_many_ranks(n):
    """`python bench.py --gpus N` must create N ranks by itself (the reference: Trainer(gpus=N,
    strategy=DDPStrategy(...)), train_egtr.py:770-779).  On CPU the launcher self-test runs under gloo: the parent
    spawns a child torch.distributed.run, rank 0 prints n_gpus and the result of a real all-reduce.  VERDICT r5: also with
    EIGHT ranks (the node size of BASELINE.json's metric), and the line carries the `train_step_ddp` object of the world > 1
    bench line (here: the bare all-reduce timer only; DESIGN.md 5 has the schema)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--launch-check"],
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["requested_gpus"] == n
    ddp = out["train_step_ddp"]
    assert ddp["allreduce_ms"] > 0 and ddp["allreduce_bytes"] > 0


def test_bench_world_gt_one_line_schema_has_the_ddp_leg():
    """The keys rank 0 adds to the JSON line for world > 1 (bench.ddp_train_leg) -- checked on the function's
    own filter with a stand-in for the train loop, since the loop itself needs GPUs."""
    import os
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    fake = {"metric": "images/sec SGG train step", "value": 800.0, "unit": "images/sec", "n_gpus": 8, "steps": 8, "warmup": 5,
            "ms_per_step": 40.0, "scaling": "weak", "dtype": "f32", "final_loss": 1.0, "rccl_ranks": 8,
            "rank_ms_per_step": {"per_rank_ms_per_step": [40.0] * 8}, "config": {"parallelism": "dp8"}, "roofline": {"x": 1}}
    orig_tb, orig_ar = bench.train_bench, bench.time_allreduce
    bench.train_bench = lambda *a, **k: dict(fake)
    bench.time_allreduce = lambda *a, **k: 2.5
    try:
        args = types.SimpleNamespace(extra_steps=8, mode="infer", batch=1, steps=50, warmup=10)
        out = bench.ddp_train_leg(args, 8, 0, torch.device("cpu"), None)
        assert bench.ddp_train_leg(args, 8, 3, torch.device("cpu"), None) is None
    finally:
        bench.train_bench, bench.time_allreduce = orig_tb, orig_ar
    for k in ("value", "ms_per_step", "n_gpus", "rccl_ranks", "rank_ms_per_step", "allreduce_ms", "allreduce_bytes",
              "allreduce_busbw_GBs", "config"):
        assert k in out, k
    assert out["allreduce_ms"] == 2.5 and "roofline" not in out
    # 165 MB over 8 ranks in 2.5 ms: bus bandwidth 2 (n - 1) / n x bytes / time
    assert abs(out["allreduce_busbw_GBs"] - 2 * 7 / 8 * 165e6 / 2.5e-3 / 1e9) < 0.01


def _compression_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from egtr_amd.runtime import DataParallelTrainer, init_distributed
    res = {}
    for mode in (None, "bf16"):
        model = _build()
        if mode is None:
            assert init_distributed() == world
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        tr = DataParallelTrainer(model, optimizer=opt, accumulate=1, clip=1e9, grad_compression=mode)
        assert tr.grad_compression == mode
        captured = {}
        orig_step = opt.step

        def step(*a, _m=model, _c=captured, _o=orig_step, **k):
            _c.update({n: p.grad.clone() for n, p in _m.named_parameters() if p.grad is not None})
            return _o(*a, **k)

        opt.step = step
        tr.training_step(_batch(100 + rank))
        res[str(mode)] = {n: g for n, g in captured.items()}
    a, b = res["None"], res["bf16"]
    worst = 0.0
    for n in a:
        scale = float(a[n].abs().max()) + 1e-30
        worst = max(worst, float((a[n] - b[n]).abs().max()) / scale)
    differs = any(not torch.equal(a[n], b[n]) for n in a)
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump({"worst_rel": worst, "differs": differs}, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bf16_gradient_compression_hook_two_ranks(tmp_path):
    """SURVEY 2b (optional, for the bandwidth-bound stress configuration): DataParallelTrainer(grad_compression="bf16")
    all-reduces 16-bit buckets.  The synchronised gradient equals the fp32 all-reduce's up to bf16 rounding of the buckets
    (8 significant bits: <= 2^-8 of each tensor's largest entry after averaging two ranks) -- and is NOT identical, i.e. the
    hook really ran."""
    world, port = 2, _free_port()
    out = str(tmp_path / "cmp")
    mp.spawn(_compression_worker, args=(world, port, out), nprocs=world, join=True)
    r = [json.load(open(f"{out}.{i}")) for i in range(world)]
    for k in range(world):
        assert r[k]["differs"] and r[k]["worst_rel"] < 2 ** -7, r[k]
