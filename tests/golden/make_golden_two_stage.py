"""Golden vectors for the two-stage branch of DeformableDetrModel.forward, produced by RUNNING THE REFERENCE
(model/deformable_detr.py:2040-2052 enc_output / pos_trans, :2075-2159 get_proposal_pos_embed /
gen_encoder_output_proposals, :2306-2337 per-token heads + top-k proposals; loss wiring model/egtr.py:459-464, 484-488,
1019-1033) on seeded weights and inputs:

    python tests/golden/make_golden_two_stage.py      -> tests/golden/sgg_small_two_stage.npz

2 images (one padded), 96 x 128, Le = 2, Ld = 3, two_stage_num_proposals = 24, with_box_refine = True (the reference's
config demands it), auxiliary losses on.  Stores the model outputs, the per-token head outputs (+inf logits kept), the
train-mode loss dict (incl. the *_enc keys) and every gradient norm."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference under the shims of _ref_import, installs the stub backbone)
import weights as W  # noqa: E402

dd, eg = MG.dd, MG.eg


def main():
    base = dict(num_queries=24, encoder_layers=2, decoder_layers=3, dropout=0.0, auxiliary_loss=True, with_box_refine=True,
                two_stage=True, two_stage_num_proposals=24)
    cfg = dd.DeformableDetrConfig(**base)
    extra = dict(num_labels=12, num_rel_labels=7, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0,
                 connectivity_loss_coefficient=30.0, smoothing=1e-14, rel_sample_negatives=80, rel_sample_nonmatching=80,
                 rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True, use_freq_bias=True,
                 use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False, logit_adj_tau=0.3,
                 output_attention_states=True)
    for k, v in extra.items():
        setattr(cfg, k, v)
    cfg_dict = {**base, **extra}
    fg = W.fg_matrix(cfg.num_labels, cfg.num_rel_labels, seed=0)
    torch.manual_seed(0)
    model = eg.DetrForSceneGraphGeneration(cfg, fg_matrix=fg)
    full = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in full.items()}
    sd = W.fill_state_dict(shapes, seed=91, alias_heads=False)
    sd["triplet_dist"], sd["rel_dist"] = full["triplet_dist"].clone(), full["rel_dist"].clone()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    rng = W.rng_inputs(92)
    B, H, Wd = 2, 96, 128
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, 72:, :] = 0
    pm[1, :, 112:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    targets = W.make_targets(93, B, cfg.two_stage_num_proposals, cfg.num_labels, cfg.num_rel_labels)
    res = dict(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=91, input_seed=92, target_seed=93, H=H, W=Wd,
               valid1=np.array([72, 112]))
    with torch.no_grad():
        out, cap, qk = MG.run_ref(model, pv, pm)
    mo = cap["mo"]
    score = mo.enc_outputs_class[..., 0]
    top = torch.sort(score, dim=1, descending=True)[0]
    margin = float((top[:, 23] - top[:, 24]).min())
    assert margin > 1e-3, f"top-k boundary too close for a parity fixture: {margin}"
    res.update(logits=MG.np_(out.logits), pred_boxes=MG.np_(out.pred_boxes), rel_mlp=MG.np_(cap["rel_mlp"]),
               conn_logits=MG.np_(cap["conn"]), inter=MG.np_(qk["inter"]), init_ref=MG.np_(qk["init_ref"]),
               inter_ref=MG.np_(mo.intermediate_reference_points), enc_outputs_class=MG.np_(mo.enc_outputs_class),
               enc_outputs_coord_logits=MG.np_(mo.enc_outputs_coord_logits), topk_margin=margin)
    model.train()
    model.zero_grad()
    out_t, _, _ = MG.run_ref(model, pv, pm, labels=targets)
    out_t.loss.backward()
    res["train_loss"] = MG.np_(out_t.loss)
    res["train_loss_dict"] = json.dumps({k: float(v) for k, v in out_t.loss_dict.items()})
    res["grad_norms"] = json.dumps({n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, "sgg_small_two_stage.npz"), **res)
    print("sgg_small_two_stage.npz", float(out_t.loss), sorted(k for k in out_t.loss_dict if k.endswith("_enc")), margin)


if __name__ == "__main__":
    main()
