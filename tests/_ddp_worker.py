"""Worker of tests/test_gpu_ddp.py (started under torch.distributed.run with ONE rank on the GPU box): the train step
through DistributedDataParallel over RCCL (backend "nccl") -- forced at world size 1 -- against the same step without
the wrapper.  The reducer's bucket hooks then run over the autograd Functions that launch HIP kernels through the C ABI
(MSDA, relation head, token linears, LayerNorm, losses), with gradient_as_bucket_view as in
egtr_amd.runtime.DataParallelTrainer (reference: Trainer(strategy=DDPStrategy(find_unused_parameters=False)),
train_egtr.py:770-779)."""
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
    from egtr_amd.runtime import DataParallelTrainer

    lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    cfg_dict = dict(num_queries=20, encoder_layers=1, decoder_layers=2, dropout=0.0, auxiliary_loss=True,
                    num_labels=12, num_rel_labels=7, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0,
                    connectivity_loss_coefficient=30.0, smoothing=1e-14, rel_sample_negatives=80,
                    rel_sample_nonmatching=80, rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True,
                    use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False,
                    logit_adj_tau=0.3)

    def build():
        torch.manual_seed(0)
        return DetrForSceneGraphGeneration(Hh.product_config(cfg_dict), fg_matrix=W.fg_matrix(12, 7)).to(dev).train()

    torch.manual_seed(1)
    batches = []
    for s in range(3):
        batches.append({"pixel_values": torch.randn(2, 3, 320, 416, device=dev),
                        "pixel_mask": torch.ones(2, 320, 416, dtype=torch.long, device=dev),
                        "labels": [{k: v.to(dev) for k, v in d.items()} for d in W.make_targets(10 + s, 2, 20, 12, 7)]})
    res = {}
    for name, force in (("ddp", True), ("plain", False)):
        model = build()
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        tr = DataParallelTrainer(model, optimizer=opt, accumulate=2, clip=0.1, force_ddp=force)
        assert (tr.model is not tr.raw) == force
        calls = {"n": 0, "bytes": 0}
        if force:
            # count the reducer's bucket all-reduces: the default hook behind a counter
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

            def counting_hook(state, bucket):
                calls["n"] += 1
                calls["bytes"] += bucket.buffer().numel() * bucket.buffer().element_size()
                return default_hooks.allreduce_hook(state, bucket)

            tr.model.register_comm_hook(None, counting_hook)
        losses, grads, per_micro = [], None, []
        for i, b in enumerate(batches + batches[:1]):       # 4 micro-steps = 2 optimizer steps (accumulate 2, no_sync)
            before = calls["n"]
            loss, _, stepped = tr.training_step(b)
            per_micro.append((calls["n"] - before, bool(stepped)))
            losses.append(float(loss))
            if i == 0:
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        res[name] = (losses, grads, {n: p.detach().clone() for n, p in model.named_parameters()})
        if force:
            grad_bytes = sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad)
            comm = {"allreduces_per_micro_step": per_micro, "allreduce_bytes_total": calls["bytes"], "grad_bytes": grad_bytes}
    (l_a, g_a, p_a), (l_b, g_b, p_b) = res["ddp"], res["plain"]
    gmax = max(float((g_a[n] - g_b[n]).abs().max()) for n in g_b)
    pmax = max(float((p_a[n] - p_b[n]).abs().max()) for n in p_b)
    gscale = max(float(g_b[n].abs().max()) for n in g_b)
    json.dump({"ranks": int(t.item()), "loss_ddp": l_a, "loss_plain": l_b, "grad_max_diff": gmax, "grad_scale": gscale, "param_max_diff": pmax,
               "n_grads": len(g_b), "grads_missing_under_ddp": sorted(set(g_b) - set(g_a)), "comm": comm}, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
