"""Generate the committed golden vectors by RUNNING THE REFERENCE ITSELF (imported from /root/reference under
the shims of _ref_import.py) on seeded inputs.  Runs only in the build container; the fixtures it writes
(tests/golden/*.npz) are data: inputs / seeds and the reference's outputs.  No reference code is copied.

    python tests/golden/make_golden.py            # regenerates every fixture

Fixtures:
  msda.npz        reference ms_deform_attn_core_pytorch (dd:925-960) fwd + autograd bwd, fp32 & fp64,
                  incl. out-of-range samples; M=8,D=32 and an odd M=3,D=20,P=2,L=2 case.
  mha.npz         DeformableDetrMultiheadAttention (dd:1107-1262): out, scaled q, k; and a run with a padding mask
                  and output_attentions=True (dd:1198-1237): out, probability map.
  sgg_small.npz   full DetrForSceneGraphGeneration (stub backbone), 2 images (one padded), N=24, Le=2, Ld=3:
                  encoder/decoder states, q/k, logits, boxes, relation logits, connectivity, gate;
                  eval-mode and train-mode loss dicts, Hungarian indices, matching costs, gradient norms.
  sgg_full.npz    600x1000, N=200, Le=Ld=6, C=150, R=50 (BASELINE config 2) with stub backbone: logits, boxes,
                  strided relation logits + checksums.
  sgg_cfg0.npz    the same for BASELINE configs[0] (N=100, 3 decoder layers); sgg_oi.npz for configs[3] (Open Images V6
                  heads: C=601, R=30).
  sgg_stress.npz  800x1333, N=300, Le=6, Ld=8, C=150, R=50 (BASELINE config 5 geometry), 2 images (one padded), stub
                  backbone, fp32 -- and the same model with weights / pixels rounded to bf16 (fp32 arithmetic): the
                  reference point for the bf16 product model.
  det_small.npz   DeformableDetrForObjectDetection + DeformableDetrLoss (dd:2400-2861), stub backbone, auxiliary losses,
                  plain and with_box_refine heads: outputs, loss dict, gradient norms.
  sgg_full_train.npz  600x1000, N=200, Le=Ld=6, bs=2, auxiliary losses ON, train mode (dropout 0): the reference's
                  loss dict, total loss and gradient norms of a few parameters.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import weights as W  # noqa: E402

dd, eg = _ref_import.load_reference()
dd.DeformableDetrTimmConvEncoder = _ref_import.make_stub_backbone_class()


def np_(t):
    return t.detach().cpu().numpy()


def ref_config(**over):
    base = dict(num_queries=24, encoder_layers=2, decoder_layers=3, dropout=0.0, auxiliary_loss=False)
    base.update({k: v for k, v in over.items() if k in ("num_queries", "encoder_layers", "decoder_layers",
                                                         "dropout", "auxiliary_loss", "with_box_refine")})
    cfg = dd.DeformableDetrConfig(**base)
    extra = dict(num_labels=12, num_rel_labels=7, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0,
                 connectivity_loss_coefficient=30.0, smoothing=1e-14, rel_sample_negatives=80,
                 rel_sample_nonmatching=80, rel_sample_negatives_largest=True,
                 rel_sample_nonmatching_largest=True, use_freq_bias=True, use_log_softmax=False,
                 freq_bias_eps=1e-12, logit_adjustment=False, logit_adj_tau=0.3, output_attention_states=True)
    extra.update({k: v for k, v in over.items() if k in extra})
    for k, v in extra.items():
        setattr(cfg, k, v)
    return cfg, {**base, **extra}


def gen_msda():
    out = {}
    cases = {
        "a": dict(B=2, Lq=29, M=8, D=32, shapes=[(9, 13), (5, 7), (3, 4), (2, 2)], P=4),
        "b": dict(B=1, Lq=11, M=3, D=20, shapes=[(6, 5), (2, 3)], P=2),
    }
    for name, c in cases.items():
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            x = W.make_msda_inputs(7 + len(name), dtype=dt, **c)
            v = x["value"].clone().requires_grad_(True)
            loc = x["loc"].clone().requires_grad_(True)
            at = x["attn"].clone().requires_grad_(True)
            o = dd.ms_deform_attn_core_pytorch(v, x["shapes"], loc, at)
            o.backward(x["grad_out"])
            out[f"{name}_{tag}_out"] = np_(o)
            out[f"{name}_{tag}_grad_value"] = np_(v.grad)
            out[f"{name}_{tag}_grad_loc"] = np_(loc.grad)
            out[f"{name}_{tag}_grad_attn"] = np_(at.grad)
        out[f"{name}_case"] = json.dumps(dict(seed=7 + len(name), **c))
    np.savez_compressed(os.path.join(HERE, "msda.npz"), **out)
    print("msda.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def gen_mha():
    torch.manual_seed(0)
    m = dd.DeformableDetrMultiheadAttention(256, 8, dropout=0.0).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = W.fill_state_dict(shapes, seed=11)
    m.load_state_dict(sd)
    rng = W.rng_inputs(12)
    x = torch.from_numpy(rng.standard_normal((2, 37, 256))).float()
    pos = torch.from_numpy(rng.standard_normal((2, 37, 256))).float()
    with torch.no_grad():
        o, _, q, k = m(x, position_embeddings=pos, output_attention_states=True)
        # the options EGTR leaves off (dd:1198-1237): a [B, N] padding mask and the probability map handed back
        mask = torch.ones(2, 37)
        mask[0, 30:] = 0
        mask[1, ::5] = 0
        om, wm, _, _ = m(x, attention_mask=mask, position_embeddings=pos, output_attentions=True)
    np.savez_compressed(os.path.join(HERE, "mha.npz"), shapes=json.dumps(shapes), seed=11, x=np_(x), pos=np_(pos),
                        out=np_(o), q=np_(q), k=np_(k), mask=np_(mask), out_masked=np_(om), attn_masked=np_(wm))
    print("mha.npz", o.shape, q.shape)


def build_ref_model(cfg_over, seed):
    cfg, cfg_dict = ref_config(**cfg_over)
    fg = W.fg_matrix(cfg.num_labels, cfg.num_rel_labels, seed=0)
    torch.manual_seed(0)
    model = eg.DetrForSceneGraphGeneration(cfg, fg_matrix=fg)
    full = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in full.items()}
    sd = W.fill_state_dict(shapes, seed=seed, alias_heads=not cfg_over.get("with_box_refine", False))
    sd["triplet_dist"], sd["rel_dist"] = full["triplet_dist"].clone(), full["rel_dist"].clone()
    # the reference's tables must equal our restated construction (egtr:169-183 precedence quirk)
    t2, r2 = W.freq_bias_tables(fg, eps=cfg.freq_bias_eps)
    assert torch.equal(t2, sd["triplet_dist"]) and torch.equal(r2, sd["rel_dist"])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return model, cfg, cfg_dict, shapes


def run_ref(model, pv, pm, labels=None):
    cap = {}
    h1 = model.rel_predictor.register_forward_hook(lambda m, i, o: cap.__setitem__("rel_mlp", o.detach().clone()))
    h2 = model.connectivity_layer.register_forward_hook(lambda m, i, o: cap.__setitem__("conn", o.detach().clone()))
    h3 = model.rel_predictor_gate.register_forward_hook(lambda m, i, o: cap.__setitem__("gate_logit", o.detach().clone()))
    h4 = model.model.register_forward_hook(lambda m, i, o: cap.__setitem__("mo", o))
    qk = {}

    def grab(m, i, o):
        qk["q"] = tuple(t.detach().clone() for t in o["decoder_attention_queries"])
        qk["k"] = tuple(t.detach().clone() for t in o["decoder_attention_keys"])
        qk["enc"] = o.encoder_last_hidden_state.detach().clone()
        qk["inter"] = o.intermediate_hidden_states.detach().clone()
        qk["init_ref"] = o.init_reference_points.detach().clone()
    h5 = model.model.register_forward_hook(grab)
    out = model(pixel_values=pv, pixel_mask=pm, labels=labels, output_attentions=False,
                output_attention_states=True, output_hidden_states=True)
    for h in (h1, h2, h3, h4, h5):
        h.remove()
    return out, cap, qk


def gen_sgg_small():
    model, cfg, cfg_dict, shapes = build_ref_model({}, seed=21)
    model.eval()
    rng = W.rng_inputs(22)
    B, H, Wd = 2, 96, 128
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, 80:, :] = 0
    pm[1, :, 104:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    targets = W.make_targets(23, B, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)
    res = dict(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=21, input_seed=22, target_seed=23,
               H=H, W=Wd, valid1=np.array([80, 104]))
    with torch.no_grad():
        out, cap, qk = run_ref(model, pv, pm)
    res.update(logits=np_(out.logits), pred_boxes=np_(out.pred_boxes), pred_rel=np_(out.pred_rel),
               pred_connectivity=np_(out.pred_connectivity), rel_mlp=np_(cap["rel_mlp"]), conn_logits=np_(cap["conn"]),
               gate_logit=np_(cap["gate_logit"]), enc=np_(qk["enc"]), inter=np_(qk["inter"]),
               init_ref=np_(qk["init_ref"]), q=np_(torch.stack(qk["q"])), k=np_(torch.stack(qk["k"])))
    # eval-mode loss (dense relation loss, egtr:806-809)
    with torch.no_grad():
        out_e, _, _ = run_ref(model, pv, pm, labels=targets)
    res["eval_loss"] = np_(out_e.loss)
    res["eval_loss_dict"] = json.dumps({k: float(v) for k, v in out_e.loss_dict.items()})
    # train-mode (dropout=0 so forward numerics are identical; criterion uses top-k sampling, egtr:798-805)
    model.train()
    model.zero_grad()
    out_t, cap_t, _ = run_ref(model, pv, pm, labels=targets)
    out_t.loss.backward()
    res["train_loss"] = np_(out_t.loss)
    res["train_loss_dict"] = json.dumps({k: float(v) for k, v in out_t.loss_dict.items()})
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    res["grad_norms"] = json.dumps(gn)
    # a few full gradient tensors (small ones) for element-wise checks
    for n in ("rel_predictor_gate.weight", "model.reference_points.weight", "class_embed.0.bias",
              "model.decoder.layers.0.encoder_attn.sampling_offsets.bias",
              "model.encoder.layers.0.self_attn.attention_weights.bias", "model.level_embed"):
        res["grad::" + n] = np_(dict(model.named_parameters())[n].grad)
    # matcher on the final-layer outputs (dd:2925-3015)
    matcher = dd.DeformableDetrHungarianMatcher(class_cost=cfg.ce_loss_coefficient, bbox_cost=cfg.bbox_cost,
                                                giou_cost=cfg.giou_cost, smoothing=cfg.smoothing)
    idx, costs = matcher({"logits": out_t.logits.detach(), "pred_boxes": out_t.pred_boxes.detach()}, targets)
    for i, ((a, b), c) in enumerate(zip(idx, costs)):
        res[f"match_pred_{i}"], res[f"match_tgt_{i}"], res[f"match_cost_{i}"] = np_(a), np_(b), np_(c)
    np.savez_compressed(os.path.join(HERE, "sgg_small.npz"), **res)
    print("sgg_small.npz", float(out_e.loss), float(out_t.loss), out.pred_rel.shape)

    # auxiliary-loss variant (egtr:1000-1017): only the loss dicts
    model2, cfg2, cfg_dict2, shapes2 = build_ref_model(dict(auxiliary_loss=True), seed=21)
    assert shapes2 == shapes
    model2.train()
    out_a, _, _ = run_ref(model2, pv, pm, labels=targets)
    np.savez_compressed(os.path.join(HERE, "sgg_small_aux.npz"), cfg=json.dumps(cfg_dict2), train_loss=np_(out_a.loss),
                        train_loss_dict=json.dumps({k: float(v) for k, v in out_a.loss_dict.items()}))
    print("sgg_small_aux.npz", float(out_a.loss))


def gen_sgg_small_refine():
    """with_box_refine=True (egtr:148-154, dd:1903-1918): per-level heads, 4-d reference points from the second
    decoder layer on (dd:1074-1081 in the cross-attention, egtr:294-295 in the box head)."""
    model, cfg, cfg_dict, shapes = build_ref_model(dict(with_box_refine=True, auxiliary_loss=True), seed=71)
    model.eval()
    rng = W.rng_inputs(72)
    B, H, Wd = 2, 96, 128
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, 72:, :] = 0
    pm[1, :, 112:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    targets = W.make_targets(73, B, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)
    res = dict(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=71, input_seed=72, target_seed=73,
               H=H, W=Wd, valid1=np.array([72, 112]))
    with torch.no_grad():
        out, cap, qk = run_ref(model, pv, pm)
    res.update(logits=np_(out.logits), pred_boxes=np_(out.pred_boxes), rel_mlp=np_(cap["rel_mlp"]),
               conn_logits=np_(cap["conn"]), inter=np_(qk["inter"]), init_ref=np_(qk["init_ref"]),
               inter_ref=np_(cap["mo"].intermediate_reference_points))
    model.train()
    model.zero_grad()
    out_t, _, _ = run_ref(model, pv, pm, labels=targets)
    out_t.loss.backward()
    res["train_loss"] = np_(out_t.loss)
    res["train_loss_dict"] = json.dumps({k: float(v) for k, v in out_t.loss_dict.items()})
    res["grad_norms"] = json.dumps({n: float(p.grad.norm()) for n, p in model.named_parameters()
                                    if p.grad is not None})
    np.savez_compressed(os.path.join(HERE, "sgg_small_refine.npz"), **res)
    print("sgg_small_refine.npz", float(out_t.loss), res["inter_ref"].shape)


def _gen_full(fname, over, seed, input_seed):
    model, cfg, cfg_dict, shapes = build_ref_model(over, seed=seed)
    model.eval()
    rng = W.rng_inputs(input_seed)
    pv = torch.from_numpy(rng.standard_normal((1, 3, 600, 1000))).float()
    pm = torch.ones(1, 600, 1000, dtype=torch.long)
    with torch.no_grad():
        out, cap, qk = run_ref(model, pv, pm)
    res = dict(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=seed, input_seed=input_seed,
               logits=np_(out.logits), pred_boxes=np_(out.pred_boxes),
               rel_mlp_strided=np_(cap["rel_mlp"][:, ::5, ::7]), conn_logits=np_(cap["conn"][..., 0]),
               pred_rel_sum=np.float64(out.pred_rel.double().sum().item()),
               pred_conn_sum=np.float64(out.pred_connectivity.double().sum().item()),
               rel_mlp_abs_sum=np.float64(cap["rel_mlp"].double().abs().sum().item()),
               enc_strided=np_(qk["enc"][:, ::37]), last_hidden=np_(qk["inter"][:, -1]))
    np.savez_compressed(os.path.join(HERE, fname), **res)
    print(fname, out.pred_rel.shape, float(res["pred_rel_sum"]))


def gen_sgg_full():
    """BASELINE configs[1] / [2] shape: Visual Genome heads, N = 200, 6 + 6 layers, one 600x1000 image."""
    _gen_full("sgg_full.npz", dict(num_queries=200, encoder_layers=6, decoder_layers=6, num_labels=150,
                                   num_rel_labels=50), 31, 32)


def gen_sgg_cfg0():
    """BASELINE configs[0]: one 600x1000 image, N = 100 queries, 3 decoder layers."""
    _gen_full("sgg_cfg0.npz", dict(num_queries=100, encoder_layers=6, decoder_layers=3, num_labels=150,
                                   num_rel_labels=50), 81, 82)


def gen_sgg_oi():
    """BASELINE configs[3]: Open Images V6 heads (601 object classes / 30 predicates), N = 200, one 600x1000 image."""
    _gen_full("sgg_oi.npz", dict(num_queries=200, encoder_layers=6, decoder_layers=6, num_labels=601,
                                 num_rel_labels=30), 91, 92)


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32) if t.is_floating_point() else t


def gen_sgg_stress():
    over = dict(num_queries=300, encoder_layers=6, decoder_layers=8, num_labels=150, num_rel_labels=50)
    model, cfg, cfg_dict, shapes = build_ref_model(over, seed=41)
    model.eval()
    rng = W.rng_inputs(42)
    B, H, Wd = 2, 800, 1333
    vh, vw = 736, 1216
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, vh:, :] = 0
    pm[1, :, vw:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    res = dict(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=41, input_seed=42, H=H, W=Wd,
               valid1=np.array([vh, vw]))

    def record(tag, pv_):
        with torch.no_grad():
            out, cap, qk = run_ref(model, pv_, pm)
        res.update({f"{tag}logits": np_(out.logits), f"{tag}pred_boxes": np_(out.pred_boxes),
                    f"{tag}rel_mlp_strided": np_(cap["rel_mlp"][:, ::5, ::7]),
                    f"{tag}conn_logits": np_(cap["conn"][..., 0]),
                    f"{tag}last_hidden": np_(qk["inter"][:, -1]),
                    f"{tag}enc_strided": np_(qk["enc"][:, ::61]),
                    f"{tag}rel_mlp_abs_sum": np.float64(cap["rel_mlp"].double().abs().sum().item())})
        print("stress", tag or "f32", out.pred_rel.shape, float(res[f"{tag}rel_mlp_abs_sum"]))

    record("", pv)
    # the bf16 reference point: identical fp32 arithmetic, weights and pixels rounded to bf16 storage precision
    sd = {k: _bf16_round(v) for k, v in model.state_dict().items()}
    keep_t, keep_r = model.state_dict()["triplet_dist"].clone(), model.state_dict()["rel_dist"].clone()
    sd["triplet_dist"], sd["rel_dist"] = keep_t, keep_r
    model.load_state_dict(sd)
    record("bf16w_", _bf16_round(pv))
    np.savez_compressed(os.path.join(HERE, "sgg_stress.npz"), **res)


def gen_sgg_full_train():
    over = dict(num_queries=200, encoder_layers=6, decoder_layers=6, num_labels=150, num_rel_labels=50,
                auxiliary_loss=True)
    model, cfg, cfg_dict, shapes = build_ref_model(over, seed=51)
    model.train()
    rng = W.rng_inputs(52)
    B, H, Wd = 2, 600, 1000
    vh, vw = 544, 928
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, vh:, :] = 0
    pm[1, :, vw:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    targets = W.make_targets(53, B, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels, tmin=5, tmax=30)
    out, cap, _ = run_ref(model, pv, pm, labels=targets)
    out.loss.backward()
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    keep = ("rel_predictor_gate.weight", "rel_predictor.layers.0.weight", "rel_predictor.layers.2.weight",
            "connectivity_layer.layers.1.weight", "class_embed.0.weight", "bbox_embed.0.layers.2.weight",
            "proj_q.0.weight", "proj_k.5.weight", "final_sub_proj.weight", "model.level_embed",
            "model.reference_points.weight", "model.query_position_embeddings.weight",
            "model.encoder.layers.0.self_attn.sampling_offsets.weight",
            "model.encoder.layers.5.self_attn.value_proj.weight",
            "model.decoder.layers.0.encoder_attn.sampling_offsets.bias",
            "model.decoder.layers.5.self_attn.q_proj.weight", "model.decoder.layers.3.fc1.weight",
            "model.input_proj.0.0.weight", "model.backbone.conv_encoder.model.2.weight")
    res = dict(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=51, input_seed=52, target_seed=53, H=H, W=Wd,
               valid1=np.array([vh, vw]), train_loss=np_(out.loss),
               train_loss_dict=json.dumps({k: float(v) for k, v in out.loss_dict.items()}),
               grad_norms=json.dumps({k: gn[k] for k in keep if k in gn}),
               logits=np_(out.logits), pred_boxes=np_(out.pred_boxes))
    res["grad::rel_predictor_gate.weight"] = np_(dict(model.named_parameters())["rel_predictor_gate.weight"].grad)
    res["grad::model.level_embed"] = np_(dict(model.named_parameters())["model.level_embed"].grad)
    np.savez_compressed(os.path.join(HERE, "sgg_full_train.npz"), **res)
    print("sgg_full_train.npz", float(out.loss), sorted(gn)[:3], len(gn))


def gen_det_small():
    """DeformableDetrForObjectDetection + DeformableDetrLoss (dd:2400-2861; what pretrain_detr.py trains): stub backbone,
    2 images (one padded), N = 24, Le = 2, Ld = 3, auxiliary losses ON -- plain heads and with_box_refine=True.
    Outputs, loss dict, total loss, every gradient norm, two full gradients."""
    res = {}
    for tag, over, seed in (("plain", dict(auxiliary_loss=True), 81), ("refine", dict(auxiliary_loss=True,
                                                                                     with_box_refine=True), 85)):
        cfg, cfg_dict = ref_config(**over)
        cfg.output_attention_states = False          # pretrain_detr.py:70
        cfg_dict["output_attention_states"] = False
        torch.manual_seed(0)
        model = dd.DeformableDetrForObjectDetection(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = W.fill_state_dict(shapes, seed=seed, alias_heads=not over.get("with_box_refine", False))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        rng = W.rng_inputs(seed + 1)
        B, H, Wd = 2, 96, 128
        pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
        pm = torch.ones(B, H, Wd, dtype=torch.long)
        pm[1, 80:, :] = 0
        pm[1, :, 104:] = 0
        pv[1] = pv[1] * pm[1][None].float()
        targets = [{k: v for k, v in t.items() if k != "rel"}
                   for t in W.make_targets(seed + 2, B, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)]
        model.eval()
        with torch.no_grad():
            out = model(pixel_values=pv, pixel_mask=pm)
        model.train()
        model.zero_grad()
        out_t = model(pixel_values=pv, pixel_mask=pm, labels=targets)
        out_t.loss.backward()
        res.update({f"{tag}_cfg": json.dumps(cfg_dict), f"{tag}_shapes": json.dumps(shapes), f"{tag}_seed": seed,
                    f"{tag}_logits": np_(out.logits), f"{tag}_pred_boxes": np_(out.pred_boxes),
                    f"{tag}_train_loss": np_(out_t.loss),
                    f"{tag}_train_loss_dict": json.dumps({k: float(v) for k, v in out_t.loss_dict.items()}),
                    f"{tag}_grad_norms": json.dumps({n: float(p.grad.norm()) for n, p in model.named_parameters()
                                                     if p.grad is not None})})
        for n in ("class_embed.0.bias", "model.reference_points.weight"):
            res[f"{tag}_grad::" + n] = np_(dict(model.named_parameters())[n].grad)
        print("det_small", tag, float(out_t.loss), len(out_t.loss_dict))
    res.update(H=96, W=128, valid1=np.array([80, 104]))
    np.savez_compressed(os.path.join(HERE, "det_small.npz"), **res)


def gen_sgg_small_edge():
    """Loss dicts, gradient norms and matcher indices of the reference for the edge-case target sets of
    weights.edge_targets() -- an image without objects, an image with one object, a crowded image -- on the model and
    inputs of sgg_small.npz."""
    res = {}
    rng = W.rng_inputs(22)
    B, H, Wd = 2, 96, 128
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, 80:, :] = 0
    pm[1, :, 104:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    for kind in ("empty", "single", "crowded"):
        model, cfg, cfg_dict, shapes = build_ref_model({}, seed=21)
        targets = W.edge_targets(kind, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)
        model.eval()
        with torch.no_grad():
            out_e, _, _ = run_ref(model, pv, pm, labels=targets)
        res[f"{kind}_eval_loss"] = np_(out_e.loss)
        res[f"{kind}_eval_loss_dict"] = json.dumps({k: float(v) for k, v in out_e.loss_dict.items()})
        model.train()
        model.zero_grad()
        out_t, _, _ = run_ref(model, pv, pm, labels=targets)
        out_t.loss.backward()
        res[f"{kind}_train_loss"] = np_(out_t.loss)
        res[f"{kind}_train_loss_dict"] = json.dumps({k: float(v) for k, v in out_t.loss_dict.items()})
        res[f"{kind}_grad_norms"] = json.dumps({n: float(p.grad.norm()) for n, p in model.named_parameters()
                                                if p.grad is not None})
        matcher = dd.DeformableDetrHungarianMatcher(class_cost=cfg.ce_loss_coefficient, bbox_cost=cfg.bbox_cost,
                                                    giou_cost=cfg.giou_cost, smoothing=cfg.smoothing)
        idx, _ = matcher({"logits": out_t.logits.detach(), "pred_boxes": out_t.pred_boxes.detach()}, targets)
        for i, (a, b) in enumerate(idx):
            res[f"{kind}_match_pred_{i}"], res[f"{kind}_match_tgt_{i}"] = np_(a), np_(b)
        print("sgg_small_edge", kind, float(out_e.loss), float(out_t.loss))
    res.update(cfg=json.dumps(cfg_dict), shapes=json.dumps(shapes), seed=21, input_seed=22, target_seed=23, H=H, W=Wd,
               valid1=np.array([80, 104]))
    np.savez_compressed(os.path.join(HERE, "sgg_small_edge.npz"), **res)


if __name__ == "__main__":
    which = sys.argv[1:] or ["msda", "mha", "small", "refine", "edge", "full", "cfg0", "oi", "stress", "full_train", "det"]
    torch.set_num_threads(8)
    if "msda" in which:
        gen_msda()
    if "mha" in which:
        gen_mha()
    if "small" in which:
        gen_sgg_small()
    if "refine" in which:
        gen_sgg_small_refine()
    if "edge" in which:
        gen_sgg_small_edge()
    if "full" in which:
        gen_sgg_full()
    if "cfg0" in which:
        gen_sgg_cfg0()
    if "oi" in which:
        gen_sgg_oi()
    if "stress" in which:
        gen_sgg_stress()
    if "full_train" in which:
        gen_sgg_full_train()
    if "det" in which:
        gen_det_small()
