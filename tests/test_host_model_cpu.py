"""CPU: the product's HOST-side logic (module API mirror, state-dict contract, glue arithmetic, matcher, loss,
autograd bridges) against the golden vectors from the reference.  The three HIP ops are replaced by
oracle-built stand-ins through the ``cpu_kernels`` fixture -- this file does NOT test the kernels."""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
import weights as W


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def small(golden_dir):
    g = Hh.load_golden(golden_dir, "sgg_small.npz")
    cfg_dict = json.loads(str(g["cfg"]))
    shapes = json.loads(str(g["shapes"]))
    return g, cfg_dict, shapes


def test_state_dict_key_contract(small):
    g, cfg_dict, shapes = small
    model, cfg, sd = Hh.build_product_model(cfg_dict, shapes, int(g["seed"]))
    own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {k: tuple(v) for k, v in shapes.items()}
    assert own == ref  # identical key set AND shapes as the instantiated reference
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    names = [n for n, _ in model.named_parameters()]
    for sub in ("backbone", "reference_points", "sampling_offsets"):  # LR grouping, train_egtr.py:427-463
        assert any(sub in n for n in names)


def test_resnet50_backbone_keys_and_shapes():
    from egtr_amd.deformable_detr import DeformableDetrConfig, DeformableDetrTimmConvEncoder
    enc = DeformableDetrTimmConvEncoder(DeformableDetrConfig())
    keys = set(enc.state_dict().keys())
    assert "model.conv1.weight" in keys and "model.layer4.2.bn3.running_var" in keys
    assert "model.layer1.0.downsample.0.weight" in keys and "model.layer2.0.downsample.1.running_mean" in keys
    assert not any("num_batches_tracked" in k for k in keys)
    n_train = sum(p.numel() for p in enc.parameters() if p.requires_grad)
    assert n_train == sum(p.numel() for n, p in enc.named_parameters() if any(f"layer{i}" in n for i in (2, 3, 4)))
    assert abs(n_train / 1e6 - 23.2) < 0.3  # SURVEY: ~23.3 M trainable backbone params
    with torch.no_grad():
        out = enc(torch.randn(1, 3, 75, 100), torch.ones(1, 75, 100, dtype=torch.long))
    assert [tuple(f.shape[1:]) for f, _ in out] == [(512, 10, 13), (1024, 5, 7), (2048, 3, 4)]


def test_forward_matches_reference(small, cpu_kernels):
    g, cfg_dict, shapes = small
    model, cfg, sd = Hh.build_product_model(cfg_dict, shapes, int(g["seed"]))
    model.load_state_dict(sd)
    model.eval()
    pv, pm = Hh.small_inputs(g)
    with torch.no_grad():
        out = model(pixel_values=pv, pixel_mask=pm, output_attentions=False, output_attention_states=True,
                    output_hidden_states=True)
    tol = 2e-4
    assert (out["logits"] - _t(g["logits"])).abs().max() < tol
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < tol
    assert (out["pred_rel"] - _t(g["pred_rel"])).abs().max() < tol
    assert (out["pred_connectivity"] - _t(g["pred_connectivity"])).abs().max() < tol
    assert (out.encoder_last_hidden_state - _t(g["enc"])).abs().max() < tol
    assert "pred_connectivity" in out and out.loss is None and "loss" not in out
    assert out.logits.shape == (2, cfg.num_queries, cfg.num_labels)
    assert len(out.decoder_hidden_states) == cfg.decoder_layers + 1


@pytest.mark.parametrize("training", [False, True])
def test_loss_and_grads_match_reference(small, cpu_kernels, training):
    import weights as W
    g, cfg_dict, shapes = small
    model, cfg, sd = Hh.build_product_model(cfg_dict, shapes, int(g["seed"]))
    model.load_state_dict(sd)
    model.train(training)
    pv, pm = Hh.small_inputs(g)
    targets = W.make_targets(int(g["target_seed"]), 2, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)
    key = "train" if training else "eval"
    with torch.set_grad_enabled(training):
        out = model(pixel_values=pv, pixel_mask=pm, labels=targets, output_attentions=False,
                    output_attention_states=True, output_hidden_states=True)
    ref = json.loads(str(g[f"{key}_loss_dict"]))
    assert set(ref) == set(out.loss_dict)
    for k, v in ref.items():
        assert abs(float(out.loss_dict[k]) - v) < 3e-4 * max(1.0, abs(v)), (k, float(out.loss_dict[k]), v)
    assert abs(float(out.loss) - float(g[f"{key}_loss"])) < 3e-4 * abs(float(g[f"{key}_loss"]))
    if training:
        out.loss.backward()
        gn = json.loads(str(g["grad_norms"]))
        params = dict(model.named_parameters())
        assert set(gn) == {n for n, p in params.items() if p.grad is not None}
        for n, v in gn.items():
            got = float(params[n].grad.norm())
            assert abs(got - v) < 2e-3 * max(abs(v), 1e-3), (n, got, v)
        for k in g.files:
            if k.startswith("grad::"):
                ref_g = _t(g[k])
                assert (params[k[6:]].grad - ref_g).abs().max() < 2e-3 * max(1.0, float(ref_g.abs().max())), k


def test_matcher_indices_bit_exact(small):
    import weights as W
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    g, cfg_dict, shapes = small
    targets = W.make_targets(int(g["target_seed"]), 2, cfg_dict["num_queries"], cfg_dict["num_labels"],
                             cfg_dict["num_rel_labels"])
    m = DeformableDetrHungarianMatcher(class_cost=cfg_dict["ce_loss_coefficient"], bbox_cost=5, giou_cost=2,
                                       smoothing=cfg_dict["smoothing"])
    idx, costs = m({"logits": _t(g["logits"]), "pred_boxes": _t(g["pred_boxes"])}, targets)
    for i, ((a, b), c) in enumerate(zip(idx, costs)):
        assert np.array_equal(a.numpy(), g[f"match_pred_{i}"]) and np.array_equal(b.numpy(), g[f"match_tgt_{i}"])
        assert np.abs(c.numpy() - g[f"match_cost_{i}"]).max() < 1e-5


@pytest.mark.parametrize("neg,nm", [(80, 80), (2, 3), (None, 5), (4, None), (0, 7)])
def test_relation_loss_device_path_equals_reference_loop(neg, nm):
    """The sync-light training path of loss_relations (masked top-k, sums / counts in un-permuted query order) against
    the line-by-line mirror of egtr:754-923 (index lists via nonzero) on the same inputs: loss_rel, loss_connectivity
    and the gradients wrt pred_rel / pred_connectivity."""
    import weights as W
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    from egtr_amd.egtr import SceneGraphGenerationLoss
    N, C, R, B = 24, 11, 6, 3
    targets = W.make_targets(21, B, N, C, R, tmin=0, tmax=7)
    targets[1]["rel"].zero_()  # an image without relations
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(B, N, C, generator=g)
    boxes = torch.rand(B, N, 4, generator=g) * 0.5 + 0.25
    matcher = DeformableDetrHungarianMatcher(class_cost=2, bbox_cost=5, giou_cost=2, smoothing=1e-14)
    indices, costs = matcher({"logits": logits, "pred_boxes": boxes}, targets)
    res = []
    for force in (False, True):
        crit = SceneGraphGenerationLoss(matcher=matcher, num_object_queries=N, num_classes=C, num_rel_labels=R,
                                        eos_coef=0.1, losses=["relations"], smoothing=1e-14, rel_sample_negatives=neg,
                                        rel_sample_nonmatching=nm, model_training=True, focal_alpha=0.25,
                                        rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True)
        crit.force_device_relations = force
        pr = torch.randn(B, N, N, R, generator=torch.Generator().manual_seed(6)).requires_grad_(True)
        pc = torch.randn(B, N, N, 1, generator=torch.Generator().manual_seed(7)).requires_grad_(True)
        out = crit.loss_relations({"pred_rel": pr, "pred_connectivity": pc}, targets, indices, costs, 1.0)
        (out["loss_rel"] * 3.0 + out["loss_connectivity"]).backward()
        res.append((float(out["loss_rel"]), float(out["loss_connectivity"]), pr.grad.clone(), pc.grad.clone()))
    (a_rel, a_conn, a_gr, a_gc), (b_rel, b_conn, b_gr, b_gc) = res
    assert abs(a_rel - b_rel) < 1e-6 * max(1.0, abs(a_rel)) and abs(a_conn - b_conn) < 1e-6
    assert (a_gr - b_gr).abs().max() < 1e-7 and (a_gc - b_gc).abs().max() < 1e-7
    assert int((a_gr != 0).sum()) == int((b_gr != 0).sum())  # the same number of selected triplets


def test_aux_loss_matches_reference(small, cpu_kernels, golden_dir):
    import weights as W
    g, cfg_dict, shapes = small
    ga = Hh.load_golden(golden_dir, "sgg_small_aux.npz")
    cfg_aux = json.loads(str(ga["cfg"]))
    model, cfg, sd = Hh.build_product_model(cfg_aux, shapes, int(g["seed"]))
    model.load_state_dict(sd)
    model.train()
    pv, pm = Hh.small_inputs(g)
    targets = W.make_targets(int(g["target_seed"]), 2, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)
    out = model(pixel_values=pv, pixel_mask=pm, labels=targets, output_attention_states=True)
    ref = json.loads(str(ga["train_loss_dict"]))
    assert set(ref) == set(out.loss_dict)
    for k, v in ref.items():
        assert abs(float(out.loss_dict[k]) - v) < 3e-4 * max(1.0, abs(v)), (k, float(out.loss_dict[k]), v)
    assert abs(float(out.loss) - float(ga["train_loss"])) < 3e-4 * abs(float(ga["train_loss"]))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No fallback: with the .so absent every kernel entry point raises."""
    import egtr_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.EgtrHipError):
        L.lib()
    from egtr_amd.load_custom import load_hip_kernels
    with pytest.raises(L.EgtrHipError):
        load_hip_kernels()


def test_output_object_and_config_roundtrip(tmp_path, small):
    from egtr_amd.deformable_detr import DeformableDetrConfig
    g, cfg_dict, shapes = small
    cfg = Hh.product_config(cfg_dict)
    cfg.save_pretrained(str(tmp_path))
    cfg2 = DeformableDetrConfig.from_pretrained(str(tmp_path))
    assert cfg2.num_queries == cfg.num_queries and cfg2.num_rel_labels == cfg.num_rel_labels
    assert cfg2.num_labels == cfg.num_labels and cfg2.smoothing == cfg.smoothing
    assert cfg2.use_return_dict and cfg2.hidden_size == 256 and cfg2.num_attention_heads == 8


def test_triplet_candidates_single_predicate_and_oi_modes_vs_reference_fixture(golden_dir):
    """egtr_amd.runtime.triplet_candidates(mode="single" / "oi") -- the other two branches of the reference's evaluate_batch
    (train_egtr.py:120-139, 154-174) -- against the reference's own outputs (postprocess_branches.npz), host tensors."""
    import helpers as Hh
    from egtr_amd.runtime import triplet_candidates
    g = Hh.load_golden(golden_dir, "postprocess_branches.npz")
    outputs, targets, meta = W.post_inputs(int(g["seed"]))
    sizes = torch.stack([t["orig_size"] for t in targets])
    single = triplet_candidates(outputs, meta["num_labels"], sizes, max_topk=100, mode="single")
    oi = triplet_candidates(outputs, meta["num_labels"], sizes, mode="oi")
    exact = [Hh.check_pred_entry({k: v.numpy() for k, v in single[j].items()}, g, j, prefix="single") for j in range(2)]
    assert exact[0] >= 99 and tuple(single[0]["rel_scores"].shape) == (100, outputs["pred_rel"].shape[-1])
    for j in range(2):
        Hh.check_oi_entry({k: v.numpy() for k, v in oi[j].items()}, g, j)
    with pytest.raises(ValueError):
        triplet_candidates(outputs, meta["num_labels"], sizes, mode="both")


def test_triplet_candidates_match_reference_postprocessing():
    """egtr_amd.runtime.triplet_candidates (batched device-side top-k) against the numpy restatement of the reference's
    evaluate_batch / argsort_desc (oracle/postprocess.py).  Clamped relation scores produce exact ties; tied triplets
    may come in either order (both are valid argsorts), everything else must agree row by row."""
    from egtr_amd.runtime import triplet_candidates
    from oracle import postprocess as OP
    g = torch.Generator().manual_seed(31)
    B, N, C, R = 3, 23, 12, 7
    outputs = {"logits": torch.randn(B, N, C + 1, generator=g) * 2, "pred_boxes": torch.rand(B, N, 4, generator=g),
               "pred_rel": torch.rand(B, N, N, R, generator=g) * 1.1 - 0.05,          # a few values outside [0, 1]
               "pred_connectivity": torch.rand(B, N, N, 1, generator=g)}
    sizes = torch.tensor([[480, 640], [600, 1000], [333, 500]])
    got = triplet_candidates(outputs, C, sizes, max_topk=100)
    for b in range(B):
        want = OP.triplet_candidates(outputs["logits"][b], outputs["pred_boxes"][b], outputs["pred_rel"][b],
                                     outputs["pred_connectivity"][b], C, sizes[b], 100)
        gi, wi = got[b]["pred_rel_inds"].numpy(), want["pred_rel_inds"]
        ts = want["triplet_scores"]
        assert np.abs(got[b]["triplet_scores"].numpy() - ts).max() < 1e-6
        uniq = np.ones(len(ts), dtype=bool)
        uniq[1:] &= ts[1:] != ts[:-1]
        uniq[:-1] &= ts[:-1] != ts[1:]
        assert uniq.sum() > 80 and np.array_equal(gi[uniq], wi[uniq])
        assert set(map(tuple, gi)) == set(map(tuple, wi))
        assert np.abs(np.sort(got[b]["rel_scores"].numpy()) - np.sort(want["rel_scores"])).max() < 1e-6
        assert np.array_equal(got[b]["pred_classes"].numpy(), want["pred_classes"])
        assert np.abs(got[b]["obj_scores"].numpy() - want["obj_scores"]).max() < 1e-6
        assert np.abs(got[b]["pred_boxes"].numpy() - want["pred_boxes"]).max() < 1e-3
    # without the connectivity head, and fewer candidates than max_topk
    small = {k_: v[:, :3, ...] if k_ in ("logits", "pred_boxes") else v[:, :3, :3] for k_, v in outputs.items()}
    small.pop("pred_connectivity")
    got = triplet_candidates(small, C, sizes, max_topk=100)
    want = OP.triplet_candidates(small["logits"][0], small["pred_boxes"][0], small["pred_rel"][0], None, C, sizes[0], 100)
    assert got[0]["pred_rel_inds"].shape == (3 * 3 * R, 3)
    nz = want["triplet_scores"] > 0   # the zero-score self pairs tie: compare the strictly positive part
    assert np.array_equal(got[0]["pred_rel_inds"].numpy()[nz], want["pred_rel_inds"][nz])


def test_bbox_overlaps_matches_reference_cython_routine():
    from egtr_amd.util import bbox_overlaps
    from oracle import postprocess as OP
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 500, (40, 2))
    a = np.concatenate([xy, xy + rng.uniform(0, 200, (40, 2))], 1)
    xy = rng.uniform(0, 500, (17, 2))
    q = np.concatenate([xy, xy + rng.uniform(0, 200, (17, 2))], 1)
    q[0] = a[0]                      # identical boxes -> 1
    q[1] = a[1] + 10000.0            # disjoint -> 0
    got = bbox_overlaps(torch.from_numpy(a), torch.from_numpy(q)).numpy()
    want = OP.bbox_overlaps(a, q)
    assert got.shape == (40, 17) and np.abs(got - want).max() < 1e-12
    assert got[0, 0] == 1.0 and got[1, 1] == 0.0



def test_postprocessing_host_path_vs_reference_fixture(golden_dir):
    """runtime.triplet_candidates and util.bbox_overlaps / bbox_intersections on host tensors against the outputs of the
    reference's evaluate_batch and Cython routines (tests/golden/postprocess.npz)."""
    import helpers as Hh
    import weights as W
    from egtr_amd.runtime import triplet_candidates
    from egtr_amd.util import bbox_intersections, bbox_overlaps
    g = Hh.load_golden(golden_dir, "postprocess.npz")
    outputs, targets, meta = W.post_inputs(int(g["seed"]))
    sizes = torch.stack([t["orig_size"] for t in targets])
    got = triplet_candidates(outputs, meta["num_labels"], sizes, max_topk=100)
    for j in range(2):
        Hh.check_pred_entry({k: v.numpy() for k, v in got[j].items()}, g, j)
    for name, (a, b) in W.bbox_cases(int(g["bbox_seed"])).items():
        assert np.abs(bbox_overlaps(torch.from_numpy(a), torch.from_numpy(b)).numpy() - g[f"iou_{name}"]).max(initial=0) < 1e-15
        assert np.abs(bbox_intersections(torch.from_numpy(a), torch.from_numpy(b)).numpy()
                      - g[f"inter_{name}"]).max(initial=0) < 1e-15


def test_cached_weights_live_on_the_owner_and_follow_their_sources():
    """derived constants must never leak from one model to the next (CPython reuses id(), the
    allocator reuses storage), and must be rebuilt when a source is modified in place or replaced."""
    import gc
    from egtr_amd import ops

    class M(torch.nn.Module):
        def __init__(self, v):
            super().__init__()
            self.w = torch.nn.Parameter(torch.full((64, 64), float(v)))

        def derived(self):
            return ops.cached_weights(self, "double", [self.w], lambda: (self.w * 2).clone())

    for trial in range(20):  # the round-1 cache returned the first model's tensor in 7 of 20 such trials
        a = M(1.0)
        assert float(a.derived()[0, 0]) == 2.0
        del a
        gc.collect()
        b = M(2.0)
        assert float(b.derived()[0, 0]) == 4.0
        del b
    m = M(3.0)
    d0 = m.derived()
    assert m.derived() is d0                      # cached
    with torch.no_grad():
        m.w.add_(1.0)                             # optimizer-style in-place update bumps the version counter
    assert float(m.derived()[0, 0]) == 8.0
    m.w = torch.nn.Parameter(torch.full((64, 64), 5.0))   # replaced parameter object
    assert float(m.derived()[0, 0]) == 10.0
    m.w.data.fill_(6.0)                           # .data edits bypass the counter: documented, explicit invalidation
    ops.invalidate_derived(m)
    assert float(m.derived()[0, 0]) == 12.0
    assert "_egtr_derived" not in m.state_dict() and not any("derived" in k for k in m.state_dict())


def test_from_pretrained_reports_key_mismatches(tmp_path):
    """a checkpoint whose keys do not match must not load silently."""
    import warnings
    from egtr_amd.deformable_detr import DeformableDetrConfig
    from egtr_amd.hf_compat import PreTrainedModel

    class Tiny(PreTrainedModel):
        config_class = DeformableDetrConfig

        def __init__(self, config):
            super().__init__(config)
            self.lin = torch.nn.Linear(4, 4)

    cfg = DeformableDetrConfig()
    cfg.save_pretrained(str(tmp_path))
    good = {"lin.weight": torch.ones(4, 4), "lin.bias": torch.zeros(4)}
    torch.save({"state_dict": {"model." + k: v for k, v in good.items()}}, str(tmp_path / "pytorch_model.bin"))
    with pytest.raises(RuntimeError, match="none of the"):
        Tiny.from_pretrained(str(tmp_path))
    torch.save({"lin.weight": torch.ones(4, 4), "extra": torch.zeros(1)}, str(tmp_path / "pytorch_model.bin"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m, info = Tiny.from_pretrained(str(tmp_path), output_loading_info=True)
    assert info["missing_keys"] == ["lin.bias"] and info["unexpected_keys"] == ["extra"]
    assert any("missing" in str(x.message) for x in w)
    assert float(m.lin.weight.sum()) == 16.0


@pytest.mark.parametrize("tag", ["plain", "refine"])
def test_object_detection_model_matches_reference(golden_dir, cpu_kernels, tag):
    """DeformableDetrForObjectDetection + DeformableDetrLoss (dd:2400-2861; imported by pretrain_detr.py:21-26) against
    the reference's own run (tests/golden/det_small.npz): state-dict contract, outputs, loss dict with auxiliary losses,
    every gradient norm.  HIP ops replaced by the oracle stand-ins: host logic only (GPU: tests/test_gpu_model.py)."""
    import weights as W
    from model.deformable_detr import DeformableDetrForObjectDetection  # the reference's import path
    g = Hh.load_golden(golden_dir, "det_small.npz")
    cfg_dict, shapes = json.loads(str(g[f"{tag}_cfg"])), json.loads(str(g[f"{tag}_shapes"]))
    seed = int(g[f"{tag}_seed"])
    model, cfg, sd = Hh.build_product_detector(cfg_dict, shapes, seed)
    assert isinstance(model, DeformableDetrForObjectDetection)
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
    model.load_state_dict(sd, strict=True)
    pv, pm = Hh.det_inputs(g, seed)
    model.eval()
    with torch.no_grad():
        out = model(pixel_values=pv, pixel_mask=pm)
    assert (out.logits - _t(g[f"{tag}_logits"])).abs().max() < 2e-4
    assert (out.pred_boxes - _t(g[f"{tag}_pred_boxes"])).abs().max() < 2e-4
    assert out.loss is None and out.auxiliary_outputs is None
    targets = [{k: v for k, v in t.items() if k != "rel"}
               for t in W.make_targets(seed + 2, 2, cfg.num_queries, cfg.num_labels, cfg.num_rel_labels)]
    model.train()
    out_t = model(pixel_values=pv, pixel_mask=pm, labels=targets)
    ref = json.loads(str(g[f"{tag}_train_loss_dict"]))
    assert set(ref) == set(out_t.loss_dict)
    for k, v in ref.items():
        assert abs(float(out_t.loss_dict[k]) - v) < 3e-4 * max(1.0, abs(v)), (k, float(out_t.loss_dict[k]), v)
    assert abs(float(out_t.loss) - float(g[f"{tag}_train_loss"])) < 3e-4 * abs(float(g[f"{tag}_train_loss"]))
    assert len(out_t.auxiliary_outputs) == cfg.decoder_layers - 1
    out_t.loss.backward()
    params = dict(model.named_parameters())
    for n, v in json.loads(str(g[f"{tag}_grad_norms"])).items():
        got = float(params[n].grad.norm())
        assert abs(got - v) < 2e-3 * max(abs(v), 1e-2), (n, got, v)
    for n in ("class_embed.0.bias", "model.reference_points.weight"):
        assert (params[n].grad - _t(g[f"{tag}_grad::" + n])).abs().max() < 1e-3 * max(1.0, float(np.abs(g[f"{tag}_grad::" + n]).max()))


def test_two_stage_model_matches_reference(golden_dir, cpu_kernels):
    """two_stage=True through the product modules (host logic; CPU stand-ins for the three HIP ops): the branch of
    DeformableDetrModel.forward at dd:2306-2337 with its helpers (:2075-2159), the extra head pair (egtr:142-163) and the
    *_enc loss terms (egtr:459-464, 484-488, 1019-1033) against the reference's own run."""
    import weights as W
    g = Hh.load_golden(golden_dir, "sgg_small_two_stage.npz")
    cfg_dict, shapes = json.loads(str(g["cfg"])), json.loads(str(g["shapes"]))
    model, cfg, _ = Hh.build_product_model(cfg_dict, shapes, int(g["seed"]))
    assert set(model.state_dict()) == set(shapes)
    sd = W.fill_state_dict(shapes, seed=int(g["seed"]), alias_heads=False)
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(W.fg_matrix(cfg.num_labels, cfg.num_rel_labels), cfg.freq_bias_eps)
    model.load_state_dict(sd)
    model.eval()
    pv, pm = Hh.small_inputs(g)
    with torch.no_grad():
        out = model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
        base = model.model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
    tol = 2e-4
    ref_box = _t(g["enc_outputs_coord_logits"])
    finite = torch.isfinite(ref_box)
    assert torch.equal(torch.isfinite(base.enc_outputs_coord_logits), finite)
    assert (base.enc_outputs_coord_logits[finite] - ref_box[finite]).abs().max() < tol
    assert (base.enc_outputs_class - _t(g["enc_outputs_class"])).abs().max() < tol
    assert (base.init_reference_points - _t(g["init_ref"])).abs().max() < tol
    assert (base.intermediate_hidden_states - _t(g["inter"])).abs().max() < tol
    assert (out.logits - _t(g["logits"])).abs().max() < tol and (out.pred_boxes - _t(g["pred_boxes"])).abs().max() < tol
    model.train()
    targets = W.make_targets(int(g["target_seed"]), 2, cfg.two_stage_num_proposals, cfg.num_labels, cfg.num_rel_labels)
    out_t = model(pixel_values=pv, pixel_mask=pm, labels=targets, output_attention_states=True)
    ref = json.loads(str(g["train_loss_dict"]))
    assert set(ref) == set(out_t.loss_dict), sorted(set(ref) ^ set(out_t.loss_dict))
    for k, v in ref.items():
        assert abs(float(out_t.loss_dict[k]) - v) < 3e-4 * max(1.0, abs(v)), (k, float(out_t.loss_dict[k]), v)
    assert abs(float(out_t.loss) - float(g["train_loss"])) < 3e-4 * abs(float(g["train_loss"]))
    out_t.loss.backward()
    gn = json.loads(str(g["grad_norms"]))
    params = dict(model.named_parameters())
    assert set(gn) == {n for n, p in params.items() if p.grad is not None}
    for n, v in gn.items():
        got = float(params[n].grad.norm())
        assert abs(got - v) < 2e-3 * max(abs(v), 1e-3), (n, got, v)


def test_fast_path_gates_announce_fall_offs_once_and_strict_mode_raises(monkeypatch):
    """egtr_amd.ops._gate / note_fallback: a predicate that turns away a call its path exists for counts it, warns ONCE per
    path, and raises under STRICT_FAST_PATH (bench.py runs strict); a call the path was never meant for is silent."""
    import warnings
    from egtr_amd import ops
    monkeypatch.setattr(ops, "FALLBACKS", {})
    monkeypatch.setattr(ops, "STRICT_FAST_PATH", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ops._gate("demo", eligible=True, ok=True, why="x") is True
        assert ops._gate("demo", eligible=False, ok=False, why="x") is False       # not eligible: silent
        assert not w and ops.FALLBACKS == {}
        assert ops._gate("demo", eligible=True, ok=False, why=lambda: "K = 100 is not a multiple of 32") is False
        assert ops._gate("demo", eligible=True, ok=False, why="again") is False
    assert ops.FALLBACKS == {"demo": 2} and len(w) == 1 and "K = 100" in str(w[0].message)
    monkeypatch.setattr(ops, "STRICT_FAST_PATH", True)
    with pytest.raises(ops.FastPathError):
        ops._gate("demo", eligible=True, ok=False, why="strict")
    # a CPU tensor is never "eligible" for the split GEMM: no fall-off recorded
    monkeypatch.setattr(ops, "FALLBACKS", {})
    assert ops.gemm_split_supported(torch.zeros(5000, 100), 150, 100) is False and ops.FALLBACKS == {}
