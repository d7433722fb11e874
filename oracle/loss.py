"""Oracle: Hungarian matcher with adaptive-smoothing offset, and the scene-graph-generation loss.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional CPU restatement of
DeformableDetrHungarianMatcher (model/deformable_detr.py:2886-3015) and SceneGraphGenerationLoss
(model/egtr.py:544-1034) for the configuration EGTR trains with: losses = labels, boxes, relations,
cardinality, uncertainty; ``rel_sample_*_largest=True`` (deterministic top-k hard negatives).

Third-party arithmetic: ``scipy.optimize.linear_sum_assignment`` (scipy 1.15.3 in the container; the
reference leaves scipy unpinned).
"""
import math

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def center_to_corners(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_iou(b1, b2):
    """model/util.py:89-102."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_box_iou(b1, b2):
    """model/util.py:105-124."""
    iou, union = box_iou(b1, b2)
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha=0.25, gamma=2):
    """model/util.py:28-59."""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def matcher_cost_matrix(logits, boxes, targets, class_cost, bbox_cost, giou_cost, smoothing):
    """dd:2946-2996. Returns cost [B, N, sum(T)] float32 (after the smoothing offset if ``smoothing``)."""
    bs, nq = logits.shape[:2]
    prob = logits.flatten(0, 1).sigmoid()
    out_bbox = boxes.flatten(0, 1)
    tgt_ids = torch.cat([t["class_labels"] for t in targets])
    tgt_bbox = torch.cat([t["boxes"] for t in targets])
    alpha, gamma = 0.25, 2.0
    neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
    pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
    c_class = pos[:, tgt_ids] - neg[:, tgt_ids]
    c_bbox = torch.cdist(out_bbox, tgt_bbox, p=1)
    c_giou = -generalized_box_iou(center_to_corners(out_bbox), center_to_corners(tgt_bbox))
    cost = bbox_cost * c_bbox + class_cost * c_class + giou_cost * c_giou
    cost = cost.view(bs, nq, -1)
    if smoothing:
        bias_eps = torch.log(torch.tensor(1e-8))
        cost_min = class_cost * (1 - alpha) * bias_eps - giou_cost  # :2990-2992
        inv_sig = -torch.log(torch.tensor((1.0 / smoothing) - 1.0))  # :2993-2995
        cost = cost - cost_min + inv_sig
    return cost


def hungarian_match(logits, boxes, targets, class_cost, bbox_cost, giou_cost, smoothing):
    """dd:2925-3015 (``@torch.no_grad()`` like the reference). Returns (indices [(i_pred, j_tgt) int64 tensors],
    matching_costs [tensors])."""
    with torch.no_grad():
        cost = matcher_cost_matrix(logits, boxes, targets, class_cost, bbox_cost, giou_cost, smoothing).cpu()
    sizes = [len(t["boxes"]) for t in targets]
    idx, costs = [], []
    for i, c in enumerate(cost.split(sizes, -1)):
        r, cidx = linear_sum_assignment(c[i])
        idx.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(cidx, dtype=torch.int64)))
        costs.append(c[i, r, cidx])
    return idx, costs


def nonmatching_cost(class_cost, bbox_cost, giou_cost, smoothing):
    """egtr:598-603."""
    return (-torch.log(torch.tensor(1e-8)) * class_cost + 4 * bbox_cost + 2 * giou_cost
            - torch.log(torch.tensor((1.0 / smoothing) - 1.0)))


def _loss_relations_one(pred_rel, target_rel, matching_cost, nm_cost, num_rel, neg, nonm, training):
    """egtr:817-923 for one image, ``*_largest=True``."""
    bce = F.binary_cross_entropy_with_logits
    if not training:  # dense branch :825-829
        w = 1.0 - matching_cost.sigmoid()
        w = torch.outer(w, w)
        return bce(pred_rel, target_rel * w.unsqueeze(-1), reduction="none").mean(-1).reshape(-1)
    matched = matching_cost != nm_cost
    n_t = int(matched.sum())
    true_idx = target_rel[:n_t, :n_t, :].nonzero()
    false_idx = (target_rel[:n_t, :n_t, :] != 1.0).nonzero()
    nonm_idx = (torch.outer(matched, matched).unsqueeze(-1).repeat(1, 1, num_rel) != True).nonzero()  # noqa: E712
    n_rel = len(true_idx)
    if neg is not None:
        if neg == 0 or n_rel == 0:
            false_idx = false_idx[[]]
        else:
            sc = pred_rel[false_idx[:, 0], false_idx[:, 1], false_idx[:, 2]]
            false_idx = false_idx[torch.topk(sc, min(n_rel * neg, sc.shape[0]), largest=True)[1]]
    if nonm is not None:
        if nonm == 0 or n_rel == 0:
            nonm_idx = nonm_idx[[]]
        else:
            sc = pred_rel[nonm_idx[:, 0], nonm_idx[:, 1], nonm_idx[:, 2]]
            nonm_idx = nonm_idx[torch.topk(sc, min(n_rel * nonm, nonm_idx.size(0)), largest=True)[1]]
    ridx = torch.cat([true_idx, false_idx, nonm_idx])
    p = pred_rel[ridx[:, 0], ridx[:, 1], ridx[:, 2]]
    t = target_rel[ridx[:, 0], ridx[:, 1], ridx[:, 2]]
    w = 1.0 - matching_cost.sigmoid()
    t = t * (w[ridx[:, 0]] * w[ridx[:, 1]])
    return bce(p, t, reduction="none")


def relation_losses(rel_logits, conn_logits, targets, indices, mcosts, nm_cost, neg, nonm, training):
    """loss_relations (egtr:754-814): per image the permutation "matched queries first" of the predictions and of the
    targets, the connectivity target, _loss_relations; loss_rel = mean over the concatenated per-image terms,
    loss_connectivity = mean over the stacked [N, N, 1] BCE maps.  rel_logits [B,N,N,R], conn_logits [B,N,N,1]
    (pre-sigmoid); indices / mcosts as the matcher returns them.  Returns (loss_rel, loss_connectivity)."""
    N, R = rel_logits.shape[1], rel_logits.shape[-1]
    rel_losses, conn_losses = [], []
    for i, ((si, ti), tgt, mc) in enumerate(zip(indices, targets, mcosts)):  # egtr:757-810
        full = torch.arange(N)
        uniq, cnt = torch.cat([full, si]).unique(return_counts=True)
        fsi = torch.cat([si, uniq[cnt == 1]])
        fti = torch.cat([ti, torch.arange(len(ti), N)])
        fmc = torch.cat([mc, torch.full((N - len(mc),), float(nm_cost), dtype=mc.dtype)])
        pr = rel_logits[i, fsi][:, fsi]
        tr = tgt["rel"][fti][:, fti]
        ri = torch.nonzero(tr)
        tconn = torch.zeros(N, N, 1, dtype=conn_logits.dtype)
        tconn[ri[:, 0], ri[:, 1]] = 1
        pc = conn_logits[i, fsi][:, fsi]
        conn_losses.append(F.binary_cross_entropy_with_logits(pc, tconn, reduction="none"))
        rel_losses.append(_loss_relations_one(pr, tr, fmc, nm_cost, R, neg, nonm, training))
    return torch.cat(rel_losses).mean(), torch.stack(conn_losses).mean()


def sgg_loss(out, targets, cfg, training):
    """SceneGraphGenerationLoss.forward (egtr:953-1034) + the weighted sum of egtr:470-494.

    ``out`` needs logits [B,N,C], pred_boxes [B,N,4], rel_logits [B,N,N,R], conn_logits [B,N,N,1]
    (pre-sigmoid, egtr:450-454) and optionally logits_all/boxes_all [B,Ld,...] for the auxiliary losses.
    Returns (total_loss, loss_dict, indices, matching_costs)."""
    cc, bc, gc, sm = cfg["ce_loss_coefficient"], cfg["bbox_cost"], cfg["giou_cost"], cfg["smoothing"]
    logits, boxes = out["logits"], out["pred_boxes"]
    B, N, C = logits.shape
    R = out["rel_logits"].shape[-1]
    indices, mcosts = hungarian_match(logits, boxes, targets, cc, bc, gc, sm)
    num_boxes = max(float(sum(len(t["class_labels"]) for t in targets)), 1.0)
    nm_cost = nonmatching_cost(cc, bc, gc, sm)
    losses = {}

    def labels_boxes_card(lg, bx, idx, suffix="", targets=targets):
        n_out = lg.shape[1]
        bidx = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(idx)])
        sidx = torch.cat([s for s, _ in idx])
        tc_o = torch.cat([t["class_labels"][j] for t, (_, j) in zip(targets, idx)])
        tc = torch.full(lg.shape[:2], C, dtype=torch.int64)
        tc[bidx, sidx] = tc_o
        onehot = torch.zeros(B, n_out, C + 1, dtype=lg.dtype)
        onehot.scatter_(2, tc.unsqueeze(-1), 1)
        losses["loss_ce" + suffix] = sigmoid_focal_loss(lg, onehot[:, :, :-1], num_boxes,
                                                        alpha=cfg["focal_alpha"], gamma=2) * n_out  # egtr:647-656
        src = bx[bidx, sidx]
        tgt = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, idx)], dim=0)
        losses["loss_bbox" + suffix] = F.l1_loss(src, tgt, reduction="none").sum() / num_boxes
        giou = 1 - torch.diag(generalized_box_iou(center_to_corners(src), center_to_corners(tgt)))
        losses["loss_giou" + suffix] = giou.sum() / num_boxes
        with torch.no_grad():
            tl = torch.as_tensor([len(t["class_labels"]) for t in targets]).float()
            cp = (lg.argmax(-1) != lg.shape[-1] - 1).sum(1).float()
            losses["cardinality_error" + suffix] = F.l1_loss(cp, tl)

    labels_boxes_card(logits, boxes, indices)

    losses["loss_rel"], losses["loss_connectivity"] = relation_losses(
        out["rel_logits"], out["conn_logits"], targets, indices, mcosts, nm_cost, cfg["rel_sample_negatives"],
        cfg["rel_sample_nonmatching"], training)

    with torch.no_grad():  # egtr:680-689
        unc = []
        for tgt, (si, ti), mc in zip(targets, indices, mcosts):
            nz = tgt["rel"][ti, :, :][:, ti, :].nonzero()
            u = mc.sigmoid()
            unc.append(u[nz[:, 0]] * u[nz[:, 1]])
        losses["uncertainty"] = torch.cat(unc).mean()

    if cfg.get("auxiliary_loss", False) and "logits_all" in out:  # egtr:1000-1017
        for li in range(out["logits_all"].shape[1] - 1):
            lg, bx = out["logits_all"][:, li], out["boxes_all"][:, li]
            aidx, _ = hungarian_match(lg, bx, targets, cc, bc, gc, sm)
            labels_boxes_card(lg, bx, aidx, suffix=f"_{li}")

    if cfg.get("two_stage", False) and out.get("enc_outputs_class") is not None:  # egtr:459-464, 1019-1033
        bin_targets = [dict(t, class_labels=torch.zeros_like(t["class_labels"])) for t in targets]
        elg, ebx = out["enc_outputs_class"], out["enc_outputs_coord_logits"].sigmoid()
        eidx, _ = hungarian_match(elg, ebx, bin_targets, cc, bc, gc, sm)
        labels_boxes_card(elg, ebx, eidx, suffix="_enc", targets=bin_targets)

    weights = {"loss_ce": cc, "loss_bbox": cfg["bbox_loss_coefficient"], "loss_giou": cfg["giou_loss_coefficient"],
               "loss_rel": cfg["rel_loss_coefficient"], "loss_connectivity": cfg["connectivity_loss_coefficient"]}
    if cfg.get("auxiliary_loss", False):
        for li in range(cfg["decoder_layers"] - 1):
            weights.update({f"{k}_{li}": v for k, v in list(weights.items()) if "_" + str(li) not in k
                            and k in ("loss_ce", "loss_bbox", "loss_giou", "loss_rel", "loss_connectivity")})
    if cfg.get("two_stage", False):  # egtr:484-488
        weights.update({f"{k}_enc": v for k, v in list(weights.items())
                        if k in ("loss_ce", "loss_bbox", "loss_giou", "loss_rel", "loss_connectivity")})
    total = sum(losses[k] * weights[k] for k in losses if k in weights)
    if "rel_gate" in out:  # egtr:496-505
        g = out["rel_gate"].reshape(B * N * N, -1).mean(0)
        for li, v in enumerate(g):
            losses[f"rel_gate_{li}"] = v
    return total, losses, indices, mcosts
