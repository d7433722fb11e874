"""One RANK of tests/test_gpu_ddp.py::test_two_ranks_real_kernels_one_gpu: two of these processes share cuda:0 and a gloo
process group (the box has ONE GPU; RCCL needs one device per rank), and run egtr_amd.runtime.DataParallelTrainer with
accumulate = 2 over the real HIP autograd nodes (5 530 token rows per micro-batch: the encoder-layer training node, the MSDA
backward with its atomics, the fused AdamW's found_inf gate).  Reference: Trainer(gpus=N, strategy=DDPStrategy(
find_unused_parameters=False), accumulate_grad_batches=2, gradient_clip_val=0.1), train_egtr.py:770-779.
Writes its findings to <out>.rank<r>.json."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]


def main(out_path):
    import helpers as Hh
    import weights as W
    from egtr_amd.egtr import DetrForSceneGraphGeneration
    from egtr_amd.runtime import DataParallelTrainer, configure_optimizers

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_dict = dict(num_queries=20, encoder_layers=1, decoder_layers=2, dropout=0.0, auxiliary_loss=True,
                    num_labels=12, num_rel_labels=7, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0,
                    connectivity_loss_coefficient=30.0, smoothing=1e-14, rel_sample_negatives=80,
                    rel_sample_nonmatching=80, rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True,
                    use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False,
                    logit_adj_tau=0.3)

    def build():
        torch.manual_seed(0)
        return DetrForSceneGraphGeneration(Hh.product_config(cfg_dict), fg_matrix=W.fg_matrix(12, 7)).to(dev).train()

    def batch(seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        return {"pixel_values": torch.randn(2, 3, 320, 416, generator=g).to(dev),
                "pixel_mask": torch.ones(2, 320, 416, dtype=torch.long, device=dev),
                "labels": [{k: v.to(dev) for k, v in d.items()} for d in W.make_targets(100 + seed, 2, 20, 12, 7)]}

    # micro-batch m of rank r: seed 10 r + m  (every rank can rebuild every batch: rank 0 also runs the global reference)
    res = {"rank": rank}
    model = build()
    opt = configure_optimizers(model, lr=1e-4, lr_backbone=1e-5, lr_initialized=None)   # AdamW, fused on the GPU
    tr = DataParallelTrainer(model, optimizer=opt, accumulate=2, clip=0.1)
    assert tr.model is not tr.raw and tr.world == 2
    calls = {"n": 0}
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

    def counting_hook(state, bucket):
        calls["n"] += 1
        return default_hooks.allreduce_hook(state, bucket)

    tr.model.register_comm_hook(None, counting_hook)
    snap = {}
    orig_step = opt.step

    def step_and_snapshot(*a, **k):   # the gradients the optimizer sees: all-reduced, accumulated, clipped
        if "grads" not in snap:
            snap["grads"] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        return orig_step(*a, **k)

    opt.step = step_and_snapshot
    per_micro = []
    for m in range(2):
        before = calls["n"]
        loss, _, stepped = tr.training_step(batch(10 * rank + m))
        per_micro.append((calls["n"] - before, bool(stepped)))
    torch.cuda.synchronize()
    res["allreduces_per_micro_step"] = per_micro
    params_after = {n: p.detach().clone() for n, p in model.named_parameters()}
    # identical weights on both ranks after the step
    sums = {n: float(p.double().sum()) for n, p in params_after.items()}
    gathered = [None, None]
    dist.all_gather_object(gathered, sums)
    res["weights_identical_across_ranks"] = gathered[0] == gathered[1]

    if rank == 0:   # the same optimizer step in ONE process on the global batch: 2 ranks x 2 micro-batches, loss / 4 each
        ref = build()
        ropt = configure_optimizers(ref, lr=1e-4, lr_backbone=1e-5, lr_initialized=None)
        for r in range(2):
            for m in range(2):
                b = batch(10 * r + m)
                out = ref(pixel_values=b["pixel_values"], pixel_mask=b["pixel_mask"], labels=b["labels"],
                          output_attentions=False, output_attention_states=True, output_hidden_states=True)
                (out.loss / 4).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        rg = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
        ropt.step()
        torch.cuda.synchronize()
        res["n_grads"] = len(rg)
        res["grads_missing_under_ddp"] = sorted(set(rg) - set(snap["grads"]))
        res["grad_scale"] = max(float(v.abs().max()) for v in rg.values())
        res["grad_max_diff"] = max(float((snap["grads"][n] - rg[n]).abs().max()) for n in rg)
        rp = dict(ref.named_parameters())
        res["param_max_diff"] = max(float((params_after[n] - rp[n].detach()).abs().max()) for n in rp)

    # ---- a cost matrix refused on ONE rank: both ranks skip the optimizer step on the device and both raise at the next step
    opt.step = orig_step
    bad = [batch(50 + 10 * rank + m) for m in range(2)]
    if rank == 1:
        bad[1]["pixel_values"][0, :, :8, :8] = float("nan")
    for b in bad:
        tr.training_step(b)
    torch.cuda.synchronize()
    res["weights_unchanged_by_refused_step"] = all(torch.equal(params_after[n], p.detach())
                                                   for n, p in model.named_parameters())
    try:
        tr.training_step(batch(90 + rank))
        res["raised_at_next_step"] = False
    except ValueError as exc:
        res["raised_at_next_step"] = str(exc)
    json.dump(res, open(f"{out_path}.rank{rank}.json", "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
