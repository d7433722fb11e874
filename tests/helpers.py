"""Shared test helpers: build the product model for a golden fixture."""
import json
import os

import numpy as np
import torch

import weights as W


def product_config(cfg_dict):
    from egtr_amd.deformable_detr import DeformableDetrConfig
    base_keys = ("num_queries", "encoder_layers", "decoder_layers", "dropout", "auxiliary_loss", "with_box_refine", "two_stage",
                 "two_stage_num_proposals")
    cfg = DeformableDetrConfig(**{k: cfg_dict[k] for k in base_keys if k in cfg_dict})
    for k, v in cfg_dict.items():
        if k not in base_keys:
            setattr(cfg, k, v)
    return cfg


def build_product_model(cfg_dict, shapes, seed, stub_backbone=True, device="cpu"):
    """Product DetrForSceneGraphGeneration with the fixture's seeded weights (and the fixtures' stub backbone)."""
    import egtr_amd.deformable_detr as pdd
    from egtr_amd.egtr import DetrForSceneGraphGeneration
    import _ref_import
    cfg = product_config(cfg_dict)
    fg = W.fg_matrix(cfg.num_labels, cfg.num_rel_labels, seed=0)
    orig = pdd.DeformableDetrTimmConvEncoder
    if stub_backbone:
        pdd.DeformableDetrTimmConvEncoder = _ref_import.make_stub_backbone_class()
    try:
        model = DetrForSceneGraphGeneration(cfg, fg_matrix=fg)
    finally:
        pdd.DeformableDetrTimmConvEncoder = orig
    sd = W.fill_state_dict(shapes, seed=seed, alias_heads=not cfg_dict.get("with_box_refine", False))
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(fg, cfg.freq_bias_eps)
    return model, cfg, sd


def small_inputs(g):
    rng = W.rng_inputs(int(g["input_seed"]))
    B, H, Wd = 2, int(g["H"]), int(g["W"])
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    vh, vw = [int(v) for v in g["valid1"]]
    pm[1, vh:, :] = 0
    pm[1, :, vw:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    return pv, pm


def load_golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def product_heads(model, pv, pm):
    """Product forward returning the PRE-sigmoid relation / connectivity logits (egtr:402-416) next to the detection
    outputs: the north-star's 1e-3 bar is stated on logits, and saturated sigmoids would hide logit error."""
    with torch.no_grad():
        outputs = model.model(pv, pixel_mask=pm, output_attentions=False, output_hidden_states=True,
                              output_attention_states=True, return_dict=True)
        enc, last = outputs.encoder_last_hidden_state, outputs.last_hidden_state
        logits, boxes, _, _, rel, conn, _, _ = model._heads(outputs, want_gate_mean=False)
    return dict(logits=logits, pred_boxes=boxes, rel_logits=rel, conn_logits=conn, last_hidden=last, enc=enc,
                inter=outputs.intermediate_hidden_states, inter_ref=outputs.intermediate_reference_points)


def rel_mlp_from_logits(rel_logits, logits, triplet_dist):
    """rel_logits - frequency bias (egtr:405-413), the bias indexed by the SAME tensor's argmax classes: what the
    reference's ``rel_predictor`` hook captured before the bias was added."""
    node = logits.argmax(-1)
    bias = torch.stack([triplet_dist[n][:, n] for n in node])
    return rel_logits - bias


def padded_inputs(g, B, dtype=torch.float32):
    rng = W.rng_inputs(int(g["input_seed"]))
    H, Wd = int(g["H"]), int(g["W"])
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    vh, vw = [int(v) for v in g["valid1"]]
    pm[1, vh:, :] = 0
    pm[1, :, vw:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    return pv.to(dtype), pm


def check_pred_entry(got, g, j, atol=1e-6, prefix="pred"):
    """``got``: one image's entry (numpy arrays) with the reference's pred_entry keys (+ optional triplet_scores);
    ``g``: tests/golden/postprocess.npz (outputs of the reference's own evaluate_batch, train_egtr.py:43-106).
    Rows are index-exact wherever the triplet score is unique; inside a group of exactly tied scores any order is a
    valid argsort (numpy's quicksort order is not a specification), so tie groups are compared as sets -- except the
    LAST group, which the top-k cut may truncate differently: there only membership in the full tie class counts."""
    want_inds, want_rel = g[f"{prefix}{j}_pred_rel_inds"], g[f"{prefix}{j}_rel_scores"]
    gi = np.asarray(got["pred_rel_inds"])
    assert gi.shape == want_inds.shape
    assert np.array_equal(np.asarray(got["pred_classes"]), g[f"{prefix}{j}_pred_classes"])
    assert np.abs(np.asarray(got["obj_scores"]) - g[f"{prefix}{j}_obj_scores"]).max() < atol
    assert np.abs(np.asarray(got["pred_boxes"]) - g[f"{prefix}{j}_pred_boxes"]).max() < 1e-3
    ts = np.asarray(got["triplet_scores"], dtype=np.float64)
    assert np.all(ts[:-1] >= ts[1:]), "triplets must come in descending score order"
    # group rows by (exactly) equal triplet score
    starts = [0] + [i for i in range(1, len(ts)) if ts[i] != ts[i - 1]] + [len(ts)]
    n_exact = 0
    for a, b in zip(starts[:-1], starts[1:]):
        last = b == len(ts)
        if b - a == 1 and not last:
            assert tuple(gi[a]) == tuple(want_inds[a]), (a, gi[a], want_inds[a])
            assert np.abs(np.asarray(got["rel_scores"][a], dtype=np.float64) - want_rel[a]).max() < atol
            n_exact += 1
        elif not last:
            assert set(map(tuple, gi[a:b])) == set(map(tuple, want_inds[a:b])), (a, b)
    return n_exact


def check_oi_entry(got, g, j, atol=1e-6):
    """One image's Open Images entry (numpy arrays, train_egtr.py:154-174) against postprocess_branches.npz."""
    ps, inds = np.asarray(got["pred_scores"]), np.asarray(got["sbj_obj_inds"])
    assert tuple(ps.shape) == tuple(g[f"oi{j}_pred_scores_shape"]) and inds.shape == (ps.shape[0], 2)
    assert np.abs(ps[::97] - g[f"oi{j}_pred_scores_strided"]).max() < atol
    assert abs(ps.astype(np.float64).sum() - float(g[f"oi{j}_pred_scores_sum"])) < 1e-6 * abs(float(g[f"oi{j}_pred_scores_sum"]))
    assert np.array_equal(inds[::97], g[f"oi{j}_sbj_obj_inds_strided"])
    assert int((inds.astype(np.int64) * np.array([1000003, 7])).sum()) == int(g[f"oi{j}_sbj_obj_inds_checksum"])
    assert np.array_equal(np.asarray(got["pred_classes"]), g[f"oi{j}_pred_classes"])
    assert np.abs(np.asarray(got["obj_scores"]) - g[f"oi{j}_obj_scores"]).max() < atol
    assert np.abs(np.asarray(got["pred_boxes"]) - g[f"oi{j}_pred_boxes"]).max() < 1e-3


def build_product_detector(cfg_dict, shapes, seed, device="cpu"):
    """Product DeformableDetrForObjectDetection with the det_small fixture's seeded weights (stub backbone)."""
    import egtr_amd.deformable_detr as pdd
    import _ref_import
    cfg = product_config(cfg_dict)
    orig = pdd.DeformableDetrTimmConvEncoder
    pdd.DeformableDetrTimmConvEncoder = _ref_import.make_stub_backbone_class()
    try:
        model = pdd.DeformableDetrForObjectDetection(cfg)
    finally:
        pdd.DeformableDetrTimmConvEncoder = orig
    sd = W.fill_state_dict(shapes, seed=seed, alias_heads=not cfg_dict.get("with_box_refine", False))
    return model, cfg, sd


def det_inputs(g, seed):
    rng = W.rng_inputs(seed + 1)
    B, H, Wd = 2, int(g["H"]), int(g["W"])
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    vh, vw = [int(v) for v in g["valid1"]]
    pm[1, vh:, :] = 0
    pm[1, :, vw:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    return pv, pm
