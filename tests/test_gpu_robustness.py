"""GPU (-m gpu): failure behaviour the reference has and a device port can silently lose.

  * a cost matrix with NaN entries: scipy raises ValueError("matrix contains invalid numeric entries") inside the
    reference's matcher (model/deformable_detr.py:3001-3005).  The device matcher must not hand -1 indices to kernels
    that dereference them: the losses skip the image, the total loss comes out NaN, and the same ValueError is raised at
    the next host synchronisation point (DeformableDetrHungarianMatcher.raise_if_invalid);
  * the "clamp iff inf / nan" backward must not modify the gradient buffer autograd handed in;
  * operands with an odd storage offset take a working path instead of EGTR_E_UNSUPPORTED.
"""
import pytest
import torch

import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _criterion(N, C, R):
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    from egtr_amd.egtr import SceneGraphGenerationLoss
    m = DeformableDetrHungarianMatcher(class_cost=2.0, bbox_cost=5.0, giou_cost=2.0, smoothing=1e-14)
    return SceneGraphGenerationLoss(
        matcher=m, num_object_queries=N, num_classes=C, num_rel_labels=R, eos_coef=0.1,
        losses=["labels", "boxes", "relations", "cardinality", "uncertainty"], smoothing=1e-14, rel_sample_negatives=80,
        rel_sample_nonmatching=80, model_training=True, focal_alpha=0.25, rel_sample_negatives_largest=True,
        rel_sample_nonmatching_largest=True).to(DEV)


@pytest.mark.parametrize("poison", [None, "logit", "box"])
def test_matcher_refusal_reaches_the_caller(poison):
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher as Matcher
    B, N, C, R = 3, 50, 12, 7
    g = torch.Generator().manual_seed(5)
    targets = [{k: v.to(DEV) for k, v in t.items()} for t in W.make_targets(3, B, N, C, R)]
    logits = torch.randn(B, N, C, generator=g).to(DEV)
    boxes = (torch.rand(B, N, 4, generator=g) * 0.5 + 0.25).to(DEV)
    if poison == "logit":
        logits[1, 7, :] = float("nan")     # (the cost matrix reads the target classes only: poison them all)
    elif poison == "box":
        boxes[2, 11, 0] = float("nan")
    out = {"logits": logits.clone().requires_grad_(True), "pred_boxes": boxes.clone().requires_grad_(True),
           "pred_rel": torch.randn(B, N, N, R, generator=g).to(DEV).requires_grad_(True),
           "pred_connectivity": torch.randn(B, N, N, 1, generator=g).to(DEV).requires_grad_(True)}
    Matcher.raise_if_invalid()                       # drain what earlier tests left behind
    crit = _criterion(N, C, R)
    losses = crit(out, targets)
    total = sum(losses[k] for k in ("loss_ce", "loss_bbox", "loss_giou", "loss_rel", "loss_connectivity"))
    total.backward()
    torch.cuda.synchronize()                           # (a fault from an out-of-bounds access would surface here)
    clean = [b for b in range(B) if not (poison == "logit" and b == 1) and not (poison == "box" and b == 2)]
    for b in clean:                                    # the other images' gradients are the usual finite ones
        assert torch.isfinite(out["pred_boxes"].grad[b]).all() and torch.isfinite(out["pred_connectivity"].grad[b]).all()
    if poison is None:
        assert torch.isfinite(total).item()
        Matcher.raise_if_invalid()                   # nothing to report
    else:
        assert torch.isnan(losses["loss_ce"]).item() and torch.isnan(total).item()
        with pytest.raises(ValueError, match="matrix contains invalid numeric entries"):
            Matcher.raise_if_invalid()
        Matcher.raise_if_invalid()                   # reported once


def test_trainer_raises_the_matchers_error_one_step_late():
    """DataParallelTrainer.training_step checks the statuses of the previous step first."""
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher as Matcher, _PENDING_MATCHER_STATUS
    Matcher.raise_if_invalid()
    host = torch.ones(2, dtype=torch.int32).pin_memory()
    ev = torch.cuda.Event()
    ev.record()
    _PENDING_MATCHER_STATUS.append((host, ev))
    from egtr_amd.runtime import DataParallelTrainer

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

    tr = DataParallelTrainer(_M().to(DEV), optimizer=torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1))
    with pytest.raises(ValueError, match="invalid numeric entries"):
        tr.training_step({})


def _tiny_train_setup():
    import helpers as Hh
    from egtr_amd.egtr import DetrForSceneGraphGeneration
    cfg = Hh.product_config(dict(num_queries=20, encoder_layers=1, decoder_layers=2, dropout=0.0, auxiliary_loss=True,
                                 num_labels=12, num_rel_labels=7, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0,
                                 connectivity_loss_coefficient=30.0, smoothing=1e-14, rel_sample_negatives=80,
                                 rel_sample_nonmatching=80, rel_sample_negatives_largest=True,
                                 rel_sample_nonmatching_largest=True, use_freq_bias=True, use_log_softmax=False,
                                 freq_bias_eps=1e-12, logit_adjustment=False, logit_adj_tau=0.3))
    torch.manual_seed(0)
    model = DetrForSceneGraphGeneration(cfg, fg_matrix=W.fg_matrix(12, 7)).to(DEV).train()
    batch = {"pixel_values": torch.randn(2, 3, 128, 160, device=DEV),
             "pixel_mask": torch.ones(2, 128, 160, dtype=torch.long, device=DEV),
             "labels": [{k: t.to(DEV) for k, t in d.items()} for d in W.make_targets(1, 2, 20, 12, 7)]}
    return model, batch


@pytest.mark.parametrize("fused", [True, False])
def test_refused_step_leaves_weights_and_optimizer_state_intact(fused):
    """a cost matrix the device matcher refuses makes loss and gradients NaN; the optimizer step of
    that window must not run.  With the fused AdamW the step is skipped ON THE DEVICE (found_inf operand) and the
    ValueError surfaces at the next step / finalize(); with any other optimizer the trainer waits for the status and
    raises before the step.  Either way parameters, AdamW moments and step counters are those of the last good step."""
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher as Matcher
    from egtr_amd.runtime import DataParallelTrainer, configure_optimizers
    Matcher.raise_if_invalid()
    Matcher.take_step_statuses()
    model, batch = _tiny_train_setup()
    opt = configure_optimizers(model, lr=1e-3, lr_backbone=1e-4, lr_initialized=None) if fused else \
        torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    tr = DataParallelTrainer(model, optimizer=opt, accumulate=1, clip=0.1)
    loss, _, stepped = tr.training_step(batch)                    # a good step: weights move
    assert stepped and torch.isfinite(loss).item()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    state = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in opt.state.items()}
    bad = dict(batch)
    bad["labels"] = [dict(t) for t in batch["labels"]]
    bad["labels"][1]["boxes"] = batch["labels"][1]["boxes"].clone()
    bad["labels"][1]["boxes"][0, 0] = float("nan")                 # -> NaN cost entries for image 1
    if fused:
        loss, _, stepped = tr.training_step(bad)                  # no host sync inside: nothing raised yet
        assert stepped and torch.isnan(loss).item()
        with pytest.raises(ValueError, match="invalid numeric entries"):
            tr.finalize()
    else:
        with pytest.raises(ValueError, match="invalid numeric entries"):
            tr.training_step(bad)
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
    for p, st in opt.state.items():
        for k, v in st.items():
            if torch.is_tensor(v):
                assert torch.equal(v, state[id(p)][k]), k
    assert all(p.grad is None for p in model.parameters())
    loss, _, _ = tr.training_step(batch)                           # and the trainer goes on with the next good batch
    tr.finalize()
    assert torch.isfinite(loss).item()
    assert any(not torch.equal(p.detach(), before[n]) for n, p in model.named_parameters())


def test_clamp_nonfinite_backward_leaves_the_incoming_gradient_alone():
    from egtr_amd import ops
    x = torch.randn(64, 256, device=DEV)
    x[3, 5] = float("inf")
    x[10, 0] = float("nan")
    xin = x.clone().requires_grad_(True)
    y = ops.clamp_nonfinite_(xin * 1.0)
    g = torch.randn(64, 256, device=DEV)
    keep = g.clone()
    y.backward(g)
    assert torch.equal(g, keep)                              # autograd's buffer was not modified
    assert float(xin.grad[3, 5]) == 0.0 and float(xin.grad[10, 0]) == 0.0
    mask = torch.ones_like(g, dtype=torch.bool)
    mask[3, 5] = mask[10, 0] = False
    assert torch.equal(xin.grad[mask], g[mask])


def test_misaligned_views_are_served():
    """A contiguous view that starts 4 bytes into its storage: column sums and the MSDA geometry pass either take an
    aligned copy or report "unsupported" to their caller's predicate -- never an exception in the step."""
    from egtr_amd import ops
    base = torch.randn(801 * 256 + 1, device=DEV)
    g = base[1:].view(801, 256)
    assert g.data_ptr() % 16 != 0 and g.is_contiguous()
    assert (ops.column_sum(g) - g.double().sum(0).float()).abs().max() < 1e-3
    off = torch.randn(2 * 40 * 384 + 1, device=DEV)[1:].view(2, 40, 384)
    ref = torch.rand(2, 40, 4, 2, device=DEV)
    assert not ops.msda_geometry_supported(off[..., :256], off[..., 256:], ref, 8, 4, 4)
    ok = torch.randn(2, 40, 384, device=DEV)
    assert ops.msda_geometry_supported(ok[..., :256], ok[..., 256:], ref, 8, 4, 4)


def test_pad_and_create_pixel_mask_on_device_equals_host_loop():
    """DeformableDetrFeatureExtractor.pad_and_create_pixel_mask with device images: one HIP launch, bit-identical to the
    host loop (pixel values and mask)."""
    from egtr_amd.feature_extraction import DeformableDetrFeatureExtractor
    fe = DeformableDetrFeatureExtractor()
    g = torch.Generator().manual_seed(2)
    imgs = [torch.randn(3, h, w, generator=g) for h, w in ((37, 61), (50, 40), (1, 1), (50, 61))]
    host = fe.pad_and_create_pixel_mask(imgs)
    dev = fe.pad_and_create_pixel_mask([x.to(DEV) for x in imgs])
    assert dev["pixel_values"].is_cuda and dev["pixel_mask"].dtype == torch.int64
    assert torch.equal(dev["pixel_values"].cpu(), host["pixel_values"])
    assert torch.equal(dev["pixel_mask"].cpu(), host["pixel_mask"])


def test_box_decode_with_class_argmax_and_shared_reference():
    """egtr_box_decode_argmax_f32: boxes identical to the plain entry; node_cls == torch.argmax of the last level's logits
    incl. exact ties (first index) and NaN (counts as the maximum); reference points expanded over the level axis (no box
    refinement) are read in place."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Ld, N, C = 2, 6, 200, 150
    delta = torch.randn(B, Ld, N, 4, generator=g).to(DEV)
    ref = torch.rand(B, N, 2, generator=g).to(DEV)
    logits = torch.randn(B, Ld, N, C, generator=g).to(DEV)
    logits[0, -1, 3, 10] = logits[0, -1, 3, 77] = 9.0            # tie: first index wins
    logits[1, -1, 5, 40] = float("nan")                          # NaN is the maximum
    logits[1, -1, 6, 0] = 50.0
    inter = ref.unsqueeze(1).expand(-1, Ld, -1, -1)              # stride 0 over the level axis
    boxes, node = ops.box_decode(delta, ref, inter, logits_all=logits)
    want = ops.box_decode(delta, ref, inter.contiguous())
    assert torch.equal(boxes, want)
    assert torch.equal(node, torch.argmax(logits[:, -1], -1))
    assert int(node[0, 3]) == 10 and int(node[1, 5]) == 40 and int(node[1, 6]) == 0
    # fewer classes than lanes, several NaNs (first one wins), a tie inside one lane's stride (c, c + 64)
    for C2 in (31, 151):
        lg = torch.randn(B, Ld, N, C2, generator=g).to(DEV)
        lg[0, -1, 8, C2 - 1] = lg[0, -1, 8, 2] = float("nan")
        lg[1, -1, 9, 5] = 30.0
        if C2 > 69:
            lg[1, -1, 9, 69] = 30.0
        _, node2 = ops.box_decode(delta, ref, inter, logits_all=lg)
        assert torch.equal(node2, torch.argmax(lg[:, -1], -1))
        assert int(node2[0, 8]) == 2 and int(node2[1, 9]) == 5
