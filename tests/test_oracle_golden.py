"""CPU: pin the oracle (oracle/) against outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py from the imported reference).  Runs anywhere, no GPU."""
import json
import os

import numpy as np
import pytest
import torch

import helpers as Hh
import weights as W
from oracle import detr as O
from oracle import loss as OL
from oracle import msda as OM


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("tag,dt,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-5)])
def test_msda_forward_backward_vs_reference(golden_dir, case, tag, dt, tol):
    g = _load(golden_dir, "msda.npz")
    c = json.loads(str(g[f"{case}_case"]))
    x = W.make_msda_inputs(c["seed"], c["B"], c["Lq"], c["M"], c["D"], [tuple(s) for s in c["shapes"]], c["P"], dtype=dt)
    out = OM.msda_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert (out - _t(g[f"{case}_{tag}_out"])).abs().max() < tol
    out_gs = OM.msda_forward_grid_sample(x["value"], x["shapes"], x["loc"], x["attn"])
    assert (out_gs - _t(g[f"{case}_{tag}_out"])).abs().max() < tol
    gv, gl, ga = OM.msda_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], x["grad_out"])
    assert (gv - _t(g[f"{case}_{tag}_grad_value"])).abs().max() < tol * 10
    assert (ga - _t(g[f"{case}_{tag}_grad_attn"])).abs().max() < tol * 10
    # grad_loc carries a factor W/H: scale tolerance
    assert (gl - _t(g[f"{case}_{tag}_grad_loc"])).abs().max() < tol * 200


def test_msda_scalar_loops_agree(golden_dir):
    g = _load(golden_dir, "msda.npz")
    c = json.loads(str(g["b_case"]))
    x = W.make_msda_inputs(c["seed"], c["B"], c["Lq"], c["M"], c["D"], [tuple(s) for s in c["shapes"]], c["P"], dtype=torch.float64)
    out = OM.msda_forward_scalar(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert (out - _t(g["b_f64_out"])).abs().max() < 1e-12


def test_msda_edge_cases():
    # every sample out of range -> zeros; exact pixel centres reproduce the pixel value
    shapes = [(4, 5), (2, 3)]
    x = W.make_msda_inputs(3, 1, 6, 2, 8, shapes, 2, dtype=torch.float64)
    loc = torch.full_like(x["loc"], 3.0)
    out = OM.msda_forward(x["value"], x["shapes"], x["lsi"], loc, x["attn"])
    assert out.abs().max() == 0
    gv, gl, ga = OM.msda_backward(x["value"], x["shapes"], x["lsi"], loc, x["attn"], x["grad_out"])
    assert gv.abs().max() == 0 and gl.abs().max() == 0 and ga.abs().max() == 0
    loc = torch.zeros_like(x["loc"])
    loc[..., 0, :, 0] = (2 + 0.5) / 5
    loc[..., 0, :, 1] = (1 + 0.5) / 4
    loc[..., 1, :, 0] = (1 + 0.5) / 3
    loc[..., 1, :, 1] = (0 + 0.5) / 2
    attn = torch.full_like(x["attn"], 0.25)
    out = OM.msda_forward(x["value"], x["shapes"], x["lsi"], loc, attn).reshape(1, 6, 2, 8)
    expect = 0.5 * x["value"][0, 1 * 5 + 2] + 0.5 * x["value"][0, 20 + 0 * 3 + 1]
    assert (out[0, 0] - expect).abs().max() < 1e-12


def test_decoder_self_attention_vs_reference(golden_dir):
    g = _load(golden_dir, "mha.npz")
    shapes = json.loads(str(g["shapes"]))
    sd = {"attn." + k: v for k, v in W.fill_state_dict(shapes, seed=int(g["seed"])).items()}
    out, q, k = O.decoder_self_attention(sd, "attn", _t(g["x"]), _t(g["pos"]))
    assert (out - _t(g["out"])).abs().max() < 2e-5
    assert (q - _t(g["q"])).abs().max() < 1e-5
    assert (k - _t(g["k"])).abs().max() < 1e-5


def _small(golden_dir):
    g = _load(golden_dir, "sgg_small.npz")
    cfg = json.loads(str(g["cfg"]))
    shapes = json.loads(str(g["shapes"]))
    sd = W.fill_state_dict(shapes, seed=int(g["seed"]))
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg["num_labels"], cfg["num_rel_labels"]), cfg["freq_bias_eps"])
    rng = W.rng_inputs(int(g["input_seed"]))
    B, H, Wd = 2, int(g["H"]), int(g["W"])
    pv = torch.from_numpy(rng.standard_normal((B, 3, H, Wd))).float()
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    vh, vw = [int(v) for v in g["valid1"]]
    pm[1, vh:, :] = 0
    pm[1, :, vw:] = 0
    pv[1] = pv[1] * pm[1][None].float()
    cfg = O_cfg(cfg)
    targets = W.make_targets(int(g["target_seed"]), B, cfg["num_queries"], cfg["num_labels"], cfg["num_rel_labels"])
    return g, cfg, sd, pv, pm, targets


def O_cfg(cfg):
    c = dict(d_model=256, num_feature_levels=4, encoder_attention_heads=8, bbox_cost=5, giou_cost=2,
             bbox_loss_coefficient=5, giou_loss_coefficient=2, focal_alpha=0.25)
    c.update(cfg)
    return c


def test_full_model_forward_vs_reference(golden_dir):
    g, cfg, sd, pv, pm, _ = _small(golden_dir)
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    tol = 2e-4
    assert (out["encoder_last_hidden_state"] - _t(g["enc"])).abs().max() < tol
    assert (out["intermediate_hidden_states"] - _t(g["inter"])).abs().max() < tol
    assert (out["init_reference_points"] - _t(g["init_ref"])).abs().max() < 1e-6
    assert (torch.stack(out["decoder_attention_queries"]) - _t(g["q"])).abs().max() < tol
    assert (torch.stack(out["decoder_attention_keys"]) - _t(g["k"])).abs().max() < tol
    assert (out["logits"] - _t(g["logits"])).abs().max() < tol
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < tol
    assert (out["conn_logits"] - _t(g["conn_logits"])).abs().max() < tol
    assert (out["pred_rel"] - _t(g["pred_rel"])).abs().max() < tol
    assert (out["pred_connectivity"] - _t(g["pred_connectivity"])).abs().max() < tol
    # pre-sigmoid relation logits = MLP output + freq bias gathered at argmax classes (egtr:405-413)
    node = out["logits"].argmax(-1)
    bias = torch.stack([sd["triplet_dist"][node[i]][:, node[i]] for i in range(2)], 0)
    assert (out["rel_logits"] - bias - _t(g["rel_mlp"])).abs().max() < tol


def test_matcher_indices_bit_exact(golden_dir):
    g, cfg, sd, pv, pm, targets = _small(golden_dir)
    idx, costs = OL.hungarian_match(_t(g["logits"]), _t(g["pred_boxes"]), targets, cfg["ce_loss_coefficient"],
                                    cfg["bbox_cost"], cfg["giou_cost"], cfg["smoothing"])
    for i, ((a, b), c) in enumerate(zip(idx, costs)):
        assert np.array_equal(a.numpy(), g[f"match_pred_{i}"])
        assert np.array_equal(b.numpy(), g[f"match_tgt_{i}"])
        assert np.abs(c.numpy() - g[f"match_cost_{i}"]).max() < 1e-5


@pytest.mark.parametrize("training", [False, True])
def test_loss_vs_reference(golden_dir, training):
    g, cfg, sd, pv, pm, targets = _small(golden_dir)
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    total, ld, _, _ = OL.sgg_loss(out, targets, cfg, training=training)
    key = "train" if training else "eval"
    ref = json.loads(str(g[f"{key}_loss_dict"]))
    assert set(ref) == set(ld), (sorted(ref), sorted(ld))
    for k, v in ref.items():
        assert abs(float(ld[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    assert abs(float(total) - float(g[f"{key}_loss"])) < 2e-4 * abs(float(g[f"{key}_loss"]))


@pytest.mark.parametrize("kind", ["empty", "single", "crowded"])
def test_loss_and_matcher_on_edge_case_targets_vs_reference(golden_dir, kind):
    """An image without objects, an image with one object and no relation, a crowded image (20 objects for 24 queries,
    dense relations): the reference's own loss dicts (eval + train) and matcher indices (sgg_small_edge.npz,
    make_golden.py edge) against the restated criterion."""
    g, cfg, sd, pv, pm, _ = _small(golden_dir)
    ge = _load(golden_dir, "sgg_small_edge.npz")
    targets = W.edge_targets(kind, cfg["num_queries"], cfg["num_labels"], cfg["num_rel_labels"])
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    for training in (False, True):
        total, ld, _, _ = OL.sgg_loss(out, targets, cfg, training=training)
        key = f"{kind}_{'train' if training else 'eval'}"
        ref = json.loads(str(ge[f"{key}_loss_dict"]))
        assert set(ref) == set(ld), (sorted(ref), sorted(ld))
        for k, v in ref.items():
            assert abs(float(ld[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
        assert abs(float(total) - float(ge[f"{key}_loss"])) < 2e-4 * abs(float(ge[f"{key}_loss"]))
    idx, _ = OL.hungarian_match(out["logits"], out["pred_boxes"], targets, cfg["ce_loss_coefficient"], cfg["bbox_cost"],
                                cfg["giou_cost"], cfg["smoothing"])
    for i, (a, b) in enumerate(idx):
        assert np.array_equal(a.numpy(), ge[f"{kind}_match_pred_{i}"])
        assert np.array_equal(b.numpy(), ge[f"{kind}_match_tgt_{i}"])


def test_aux_loss_vs_reference(golden_dir):
    g, cfg, sd, pv, pm, targets = _small(golden_dir)
    ga = _load(golden_dir, "sgg_small_aux.npz")
    cfg = O_cfg(json.loads(str(ga["cfg"])))
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    total, ld, _, _ = OL.sgg_loss(out, targets, cfg, training=True)
    ref = json.loads(str(ga["train_loss_dict"]))
    assert set(ref) == set(ld), (sorted(set(ref) ^ set(ld)))
    for k, v in ref.items():
        assert abs(float(ld[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    assert abs(float(total) - float(ga["train_loss"])) < 2e-4 * abs(float(ga["train_loss"]))


def test_box_refine_forward_and_loss_vs_reference(golden_dir):
    """with_box_refine=True (egtr:148-154): per-level heads, decoder-side refinement (dd:1903-1918), 4-d reference
    points in the cross-attention (dd:1074-1081) and in the box head (egtr:294-295)."""
    g = _load(golden_dir, "sgg_small_refine.npz")
    cfg = O_cfg(json.loads(str(g["cfg"])))
    sd = W.fill_state_dict(json.loads(str(g["shapes"])), seed=int(g["seed"]), alias_heads=False)
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg["num_labels"], cfg["num_rel_labels"]),
                                                            cfg["freq_bias_eps"])
    pv, pm = Hh.small_inputs(g)
    targets = W.make_targets(int(g["target_seed"]), 2, cfg["num_queries"], cfg["num_labels"], cfg["num_rel_labels"])
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    tol = 2e-4
    assert out["intermediate_reference_points"].shape[-1] == 4
    assert (out["intermediate_reference_points"] - _t(g["inter_ref"])).abs().max() < tol
    assert (out["intermediate_hidden_states"] - _t(g["inter"])).abs().max() < tol
    assert (out["logits"] - _t(g["logits"])).abs().max() < tol
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < tol
    assert (out["conn_logits"] - _t(g["conn_logits"])).abs().max() < tol
    node = out["logits"].argmax(-1)
    bias = torch.stack([sd["triplet_dist"][node[i]][:, node[i]] for i in range(2)], 0)
    assert (out["rel_logits"] - bias - _t(g["rel_mlp"])).abs().max() < tol
    total, ld, _, _ = OL.sgg_loss(out, targets, cfg, training=True)
    ref = json.loads(str(g["train_loss_dict"]))
    assert set(ref) == set(ld), sorted(set(ref) ^ set(ld))
    for k, v in ref.items():
        assert abs(float(ld[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    assert abs(float(total) - float(g["train_loss"])) < 2e-4 * abs(float(g["train_loss"]))


def test_two_stage_forward_and_loss_vs_reference(golden_dir):
    """two_stage=True (dd:2040-2052, 2075-2159, 2306-2337; egtr:459-464, 484-488, 1019-1033): per-token proposal heads,
    top-k reference boxes, pos_trans queries, the *_enc loss terms -- the oracle against the reference's own run."""
    g = _load(golden_dir, "sgg_small_two_stage.npz")
    cfg = O_cfg(json.loads(str(g["cfg"])))
    sd = W.fill_state_dict(json.loads(str(g["shapes"])), seed=int(g["seed"]), alias_heads=False)
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg["num_labels"], cfg["num_rel_labels"]),
                                                            cfg["freq_bias_eps"])
    pv, pm = Hh.small_inputs(g)
    K = cfg["two_stage_num_proposals"]
    targets = W.make_targets(int(g["target_seed"]), 2, K, cfg["num_labels"], cfg["num_rel_labels"])
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    tol = 2e-4
    ref_box = _t(g["enc_outputs_coord_logits"])
    finite = torch.isfinite(ref_box)
    assert torch.equal(torch.isfinite(out["enc_outputs_coord_logits"]), finite) and not finite.all()
    assert (out["enc_outputs_coord_logits"][finite] - ref_box[finite]).abs().max() < tol
    assert (out["enc_outputs_class"] - _t(g["enc_outputs_class"])).abs().max() < tol
    assert out["init_reference_points"].shape == (2, K, 4)
    assert (out["init_reference_points"] - _t(g["init_ref"])).abs().max() < tol
    assert (out["intermediate_hidden_states"] - _t(g["inter"])).abs().max() < tol
    assert (out["logits"] - _t(g["logits"])).abs().max() < tol
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < tol
    assert (out["conn_logits"] - _t(g["conn_logits"])).abs().max() < tol
    total, ld, _, _ = OL.sgg_loss(out, targets, cfg, training=True)
    ref = json.loads(str(g["train_loss_dict"]))
    assert set(ref) == set(ld), sorted(set(ref) ^ set(ld))
    assert {"loss_ce_enc", "loss_bbox_enc", "loss_giou_enc", "cardinality_error_enc"} <= set(ld)
    for k, v in ref.items():
        assert abs(float(ld[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    assert abs(float(total) - float(g["train_loss"])) < 2e-4 * abs(float(g["train_loss"]))


def full_case(golden_dir, fixture="sgg_full.npz"):
    g = _load(golden_dir, fixture)
    cfg = O_cfg(json.loads(str(g["cfg"])))
    shapes = json.loads(str(g["shapes"]))
    sd = W.fill_state_dict(shapes, seed=int(g["seed"]))
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg["num_labels"], cfg["num_rel_labels"]), cfg["freq_bias_eps"])
    rng = W.rng_inputs(int(g["input_seed"]))
    pv = torch.from_numpy(rng.standard_normal((1, 3, 600, 1000))).float()
    pm = torch.ones(1, 600, 1000, dtype=torch.long)
    return g, cfg, sd, pv, pm


@pytest.mark.parametrize("fixture", ["sgg_full.npz", "sgg_cfg0.npz", "sgg_oi.npz"])
def test_full_size_600x1000_vs_reference(golden_dir, fixture):
    """One 600x1000 image with the stub backbone at BASELINE configs[1]/[2] (N=200, 6 + 6 layers, C=150, R=50),
    configs[0] (N=100, 3 decoder layers) and configs[3] (Open Images V6 heads: C=601, R=30)."""
    g, cfg, sd, pv, pm = full_case(golden_dir, fixture)
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
    tol = 5e-4
    assert (out["logits"] - _t(g["logits"])).abs().max() < tol
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < tol
    assert (out["last_hidden_state"] - _t(g["last_hidden"])).abs().max() < tol
    assert (out["encoder_last_hidden_state"][:, ::37] - _t(g["enc_strided"])).abs().max() < tol
    node = out["logits"].argmax(-1)
    bias = sd["triplet_dist"][node[0]][:, node[0]][None]
    assert ((out["rel_logits"] - bias)[:, ::5, ::7] - _t(g["rel_mlp_strided"])).abs().max() < tol
    assert (out["conn_logits"][..., 0] - _t(g["conn_logits"])).abs().max() < tol
    assert abs(out["pred_rel"].double().sum().item() - float(g["pred_rel_sum"])) < 1.0


def O_cfg_of(g):
    return O_cfg(json.loads(str(g["cfg"])))


def test_stress_geometry_800x1333_vs_reference(golden_dir):
    """BASELINE configs[4] geometry (800x1333, N=300, 6 enc / 8 dec; image 1 padded), fp32 and the bf16-rounded-weights
    variant: pins the oracle where the stress GPU tests use it."""
    import helpers as Hh
    g = _load(golden_dir, "sgg_stress.npz")
    cfg = O_cfg_of(g)
    shapes = json.loads(str(g["shapes"]))
    sd = W.fill_state_dict(shapes, seed=int(g["seed"]))
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg["num_labels"], cfg["num_rel_labels"]),
                                                            cfg["freq_bias_eps"])
    pv, pm = Hh.padded_inputs(g, 2)
    # image 0 only (keeps the CPU suite short); the padded image is covered on the GPU against the same fixture
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv[:1], pm[:1])
    tol = 5e-4
    assert (out["logits"] - _t(g["logits"])[:1]).abs().max() < tol
    assert (out["pred_boxes"] - _t(g["pred_boxes"])[:1]).abs().max() < tol
    rel_mlp = Hh.rel_mlp_from_logits(out["rel_logits"], out["logits"], sd["triplet_dist"])
    assert (rel_mlp[:, ::5, ::7] - _t(g["rel_mlp_strided"])[:1]).abs().max() < tol
    assert (out["conn_logits"][..., 0] - _t(g["conn_logits"])[:1]).abs().max() < tol


def test_full_size_train_aux_loss_vs_reference(golden_dir):
    """600x1000, bs=2 (one padded), N=200, auxiliary losses on, train-mode criterion: every loss-dict entry."""
    import helpers as Hh
    g = _load(golden_dir, "sgg_full_train.npz")
    cfg = O_cfg_of(g)
    shapes = json.loads(str(g["shapes"]))
    sd = W.fill_state_dict(shapes, seed=int(g["seed"]))
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg["num_labels"], cfg["num_rel_labels"]),
                                                            cfg["freq_bias_eps"])
    pv, pm = Hh.padded_inputs(g, 2)
    targets = W.make_targets(int(g["target_seed"]), 2, cfg["num_queries"], cfg["num_labels"], cfg["num_rel_labels"],
                             tmin=5, tmax=30)
    with torch.no_grad():
        out = O.sgg_forward(sd, cfg, pv, pm)
        total, ld, _, _ = OL.sgg_loss(out, targets, cfg, training=True)
    ref = json.loads(str(g["train_loss_dict"]))
    assert set(ref) == set(ld), sorted(set(ref) ^ set(ld))
    for k, v in ref.items():
        assert abs(float(ld[k]) - v) < 3e-4 * max(1.0, abs(v)), (k, float(ld[k]), v)
    assert abs(float(total) - float(g["train_loss"])) < 3e-4 * abs(float(g["train_loss"]))


def test_postprocessing_oracle_vs_reference_evaluate_batch(golden_dir):
    """oracle/postprocess.py pinned against the reference's own evaluate_batch (imported and run by
    tests/golden/make_golden_post.py) and its Cython bbox routines (compiled by oracle/Makefile): postprocess.npz."""
    import helpers as Hh
    from oracle import postprocess as OP
    g = _load(golden_dir, "postprocess.npz")
    outputs, targets, meta = W.post_inputs(int(g["seed"]))
    exact = []
    for j in range(2):
        got = OP.triplet_candidates(outputs["logits"][j], outputs["pred_boxes"][j], outputs["pred_rel"][j],
                                    outputs["pred_connectivity"][j], meta["num_labels"], targets[j]["orig_size"], 100)
        exact.append(Hh.check_pred_entry(got, g, j))
        if j == 0:  # continuous scores: the restatement reproduces numpy's order row for row
            assert np.array_equal(got["pred_rel_inds"], g["pred0_pred_rel_inds"])
            assert np.array_equal(got["rel_scores"], g["pred0_rel_scores"])
    assert exact[0] >= 99 and exact[1] < 99   # image 1 was built with tied scores
    for name, (a, b) in W.bbox_cases(int(g["bbox_seed"])).items():
        assert np.array_equal(OP.bbox_overlaps(a, b), g[f"iou_{name}"]), name      # bit-exact float64


def test_postprocessing_oracle_single_predicate_and_oi_branches_vs_reference(golden_dir):
    """oracle.postprocess.pair_candidates / oi_candidates pinned against the reference's own evaluate_batch run with the
    single-predicate and Open Images evaluators (tests/golden/make_golden_post_branches.py -> postprocess_branches.npz)."""
    import helpers as Hh
    from oracle import postprocess as OP
    g = _load(golden_dir, "postprocess_branches.npz")
    outputs, targets, meta = W.post_inputs(int(g["seed"]))
    for j in range(2):
        args = (outputs["logits"][j], outputs["pred_boxes"][j], outputs["pred_rel"][j], outputs["pred_connectivity"][j],
                meta["num_labels"], targets[j]["orig_size"])
        got = OP.pair_candidates(*args, 100)
        n_exact = Hh.check_pred_entry(got, g, j, prefix="single")
        assert got["rel_scores"].shape == (100, outputs["pred_rel"].shape[-1])
        if j == 0:
            assert n_exact >= 99 and np.array_equal(got["pred_rel_inds"], g["single0_pred_rel_inds"])
            assert np.array_equal(got["rel_scores"], g["single0_rel_scores"])
        Hh.check_oi_entry(OP.oi_candidates(*args), g, j)


def test_oracle_ref_bbox_module_when_built(golden_dir):
    """oracle/_ref (the reference's Cython source compiled here) reproduces the committed fixture; skipped where the
    module was not built (a checkout without /root/reference)."""
    from oracle import ref_bbox
    m = ref_bbox.load()
    if m is None:
        pytest.skip("oracle/_ref/bbox*.so not built")
    g = _load(golden_dir, "postprocess.npz")
    for name, (a, b) in W.bbox_cases(int(g["bbox_seed"])).items():
        assert np.array_equal(m.bbox_overlaps(a, b), g[f"iou_{name}"])
        assert np.array_equal(m.bbox_intersections(a, b), g[f"inter_{name}"])


def test_lsa_restatement_equals_scipy(golden_dir):
    """oracle/lsa.py (the restated algorithm of scipy.optimize.linear_sum_assignment, the third-party solver behind the
    reference's matcher, scipy 1.15.3 here) against scipy itself, index for index: random, tie-heavy integer, constant,
    wide / tall / square / empty matrices, and the matcher's own cost matrix shapes."""
    from scipy.optimize import linear_sum_assignment as sp
    from oracle.lsa import linear_sum_assignment as mine
    rng = np.random.default_rng(0)
    for trial in range(1500):
        kind = trial % 6
        nr, nc = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if kind == 0:
            c = rng.standard_normal((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif kind == 2:
            c = np.full((nr, nc), 1.5)
        elif kind == 3:
            c = rng.integers(0, 2, (nr, nc)).astype(float)
        elif kind == 4:
            c = rng.standard_normal((nr, nc)).astype(np.float32).astype(float)
        else:
            c = np.round(rng.standard_normal((nr, nc)) * 2) / 2
        a, b = sp(c)
        a2, b2 = mine(c)
        assert np.array_equal(a, a2) and np.array_equal(b, b2), (trial, kind, nr, nc)
    for T in (1, 5, 30, 62, 200, 230):
        c = rng.standard_normal((200, T)).astype(np.float32)
        for cc in (c, np.round(c)):
            a, b = sp(cc)
            a2, b2 = mine(cc)
            assert np.array_equal(a, a2) and np.array_equal(b, b2), T
    assert mine(np.zeros((0, 5)))[0].size == 0
    with pytest.raises(ValueError):
        mine(np.array([[np.nan, 1.0]]))
