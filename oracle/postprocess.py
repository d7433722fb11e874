"""ORACLE (test infrastructure only -- never imported by egtr_amd/): CPU / numpy restatement of the reference's
scene-graph post-processing, the inputs of its evaluators.

    triplet_candidates  <- evaluate_batch, train_egtr.py:54-106 (multiple-predicate branch) with argsort_desc,
                           lib/pytorch_misc.py:27-34, and rescale_bboxes, util/box_ops.py:87-91
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
