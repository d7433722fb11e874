"""ORACLE (test infrastructure only -- never imported by egtr_amd/): CPU / numpy restatement of the reference's
scene-graph post-processing, the inputs of its evaluators.

    triplet_candidates  <- evaluate_batch, train_egtr.py:54-106 (multiple-predicate branch) with argsort_desc,
                           lib/pytorch_misc.py:27-34, and rescale_bboxes, util/box_ops.py:87-91
    pair_candidates     <- evaluate_batch, train_egtr.py:120-139 (single-predicate branch)
    oi_candidates       <- evaluate_batch, train_egtr.py:154-174 (Open Images branch: every pair)
    bbox_overlaps       <- lib/fpn/box_intersections_cpu/bbox.pyx:21-61 (Cython in the reference; loops restated)

PINNED: tests/test_oracle_golden.py::test_postprocessing_oracle_vs_reference_evaluate_batch checks both functions
against tests/golden/postprocess.npz (and postprocess_branches.npz for the single-predicate / Open Images branches), i.e. against the outputs of the reference's own evaluate_batch (imported from
/root/reference and run by tests/golden/make_golden_post.py) and of its Cython routines compiled from the reference
sources (oracle/Makefile -> oracle/_ref/).
"""
import numpy as np
import torch


def triplet_candidates(logits, pred_boxes, pred_rel, pred_connectivity, num_labels, orig_size, max_topk=100):
    """One image.  logits [N, C+1], pred_boxes [N, 4] cxcywh in [0, 1], pred_rel [N, N, R] / pred_connectivity
    [N, N, 1] post-sigmoid, orig_size (h, w).  Returns the reference's pred_entry dict (numpy)."""
    obj_scores, pred_classes = torch.max(logits.softmax(-1)[:, :num_labels], -1)      # train_egtr.py:57-59
    sub_ob = torch.outer(obj_scores, obj_scores)
    n = logits.size(0)
    sub_ob[torch.arange(n), torch.arange(n)] = 0.0                                     # :61-63 no self-connection
    rel = torch.clamp(pred_rel, 0.0, 1.0)                                              # :66
    if pred_connectivity is not None:
        rel = rel * torch.clamp(pred_connectivity, 0.0, 1.0)                           # :67-69
    scores = (rel * sub_ob.unsqueeze(-1)).numpy()                                      # :86
    order = np.argsort(-scores.ravel())                                                # pytorch_misc.py:34
    inds = np.column_stack(np.unravel_index(order, scores.shape))[:max_topk]           # :87-89
    rel_np = rel.numpy()
    rel_scores = rel_np[inds[:, 0], inds[:, 1], inds[:, 2]]                            # :90-94
    h, w = float(orig_size[0]), float(orig_size[1])
    cx, cy, bw, bh = pred_boxes.unbind(-1)                                             # box_ops.py:87-91
    xyxy = torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], -1)
    boxes = (xyxy * torch.tensor([w, h, w, h], dtype=torch.float32)).numpy()
    return {"pred_boxes": boxes, "pred_classes": pred_classes.numpy(), "obj_scores": obj_scores.numpy(),
            "pred_rel_inds": inds, "rel_scores": rel_scores, "triplet_scores": scores[inds[:, 0], inds[:, 1], inds[:, 2]]}


def _common(logits, pred_boxes, pred_rel, pred_connectivity, num_labels, orig_size):
    obj_scores, pred_classes = torch.max(logits.softmax(-1)[:, :num_labels], -1)      # train_egtr.py:57-59
    sub_ob = torch.outer(obj_scores, obj_scores)
    n = logits.size(0)
    sub_ob[torch.arange(n), torch.arange(n)] = 0.0                                     # :61-63
    rel = torch.clamp(pred_rel, 0.0, 1.0)                                              # :66
    if pred_connectivity is not None:
        rel = rel * torch.clamp(pred_connectivity, 0.0, 1.0)                           # :67-69
    h, w = float(orig_size[0]), float(orig_size[1])
    cx, cy, bw, bh = pred_boxes.unbind(-1)                                             # box_ops.py:87-91
    xyxy = torch.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], -1)
    boxes = (xyxy * torch.tensor([w, h, w, h], dtype=torch.float32)).numpy()
    return obj_scores, pred_classes, sub_ob, rel, boxes


def pair_candidates(logits, pred_boxes, pred_rel, pred_connectivity, num_labels, orig_size, max_topk=100):
    """train_egtr.py:120-139 (single-predicate evaluator): the best (subject, object) pairs by
    max_p(pred_rel) * score_s * score_o; rel_scores = the pair's whole predicate row [k, R]."""
    obj_scores, pred_classes, sub_ob, rel, boxes = _common(logits, pred_boxes, pred_rel, pred_connectivity, num_labels,
                                                           orig_size)
    scores = (rel.max(-1)[0] * sub_ob).numpy()                                         # :121
    order = np.argsort(-scores.ravel())                                                # pytorch_misc.py:34
    inds = np.column_stack(np.unravel_index(order, scores.shape))[:max_topk]           # :122-124
    rel_scores = rel.numpy()[inds[:, 0], inds[:, 1]]                                   # :125-127
    return {"pred_boxes": boxes, "pred_classes": pred_classes.numpy(), "obj_scores": obj_scores.numpy(),
            "pred_rel_inds": inds, "rel_scores": rel_scores, "triplet_scores": scores[inds[:, 0], inds[:, 1]]}


def oi_candidates(logits, pred_boxes, pred_rel, pred_connectivity, num_labels, orig_size):
    """train_egtr.py:154-174 (Open Images evaluator): every (subject, object) pair with its predicate row."""
    obj_scores, pred_classes, _, rel, boxes = _common(logits, pred_boxes, pred_rel, pred_connectivity, num_labels,
                                                      orig_size)
    n = logits.size(0)
    pairs = torch.cartesian_prod(torch.arange(n), torch.arange(n)).numpy()            # :155-157
    return {"pred_boxes": boxes, "pred_classes": pred_classes.numpy(), "obj_scores": obj_scores.numpy(),
            "sbj_obj_inds": pairs, "pred_scores": rel.numpy().reshape(-1, rel.size(-1))}   # :158-160


def bbox_overlaps(boxes, query_boxes):
    """bbox.pyx:21-61, loop for loop (float64, "+1 pixel" convention)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    query = np.ascontiguousarray(query_boxes, dtype=np.float64)
    n_, k_ = boxes.shape[0], query.shape[0]
    out = np.zeros((n_, k_), dtype=np.float64)
    for k in range(k_):
        box_area = (query[k, 2] - query[k, 0] + 1) * (query[k, 3] - query[k, 1] + 1)
        for n in range(n_):
            iw = min(boxes[n, 2], query[k, 2]) - max(boxes[n, 0], query[k, 0]) + 1
            if iw > 0:
                ih = min(boxes[n, 3], query[k, 3]) - max(boxes[n, 1], query[k, 1]) + 1
                if ih > 0:
                    ua = float((boxes[n, 2] - boxes[n, 0] + 1) * (boxes[n, 3] - boxes[n, 1] + 1) + box_area - iw * ih)
                    out[n, k] = iw * ih / ua
    return out

