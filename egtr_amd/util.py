"""Loss / box helpers used by the matcher and the SGG loss -- host-side PyTorch, mirroring the reference's
model/util.py (sigmoid_focal_loss :28-59, box_iou :89-102, generalized_box_iou :105-124, dice_loss :9-25,
NestedTensor :139-175) so that train_egtr.py-style callers find the same names."""
from typing import List, Optional

import torch
from torch import Tensor, nn


def center_to_corners_format(x):
    """(cx, cy, w, h) -> (x0, y0, x1, y1); transformers' helper the reference imports at model/egtr.py:35."""
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def dice_loss(inputs, targets, num_boxes):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    return (1 - (numerator + 1) / (denominator + 1)).sum() / num_boxes


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    prob = inputs.sigmoid()
    ce_loss = nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def _upcast(t: Tensor) -> Tensor:
    if t.is_floating_point():
        return t if t.dtype in (torch.float32, torch.float64) else t.float()
    return t if t.dtype in (torch.int32, torch.int64) else t.int()


def box_area(boxes: Tensor) -> Tensor:
    boxes = _upcast(boxes)
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1, boxes2):
    area1, area2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / union, union


def generalized_box_iou(boxes1, boxes2):
    """Boxes in corner format; returns the [N, M] pairwise GIoU matrix."""
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), self.mask.to(device) if self.mask is not None else None)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]):
    if tensor_list[0].ndim != 3:
        raise ValueError("Only 3-dimensional tensors are supported")
    max_size = [max(s) for s in zip(*[list(img.shape) for img in tensor_list])]
    b, (c, h, w) = len(tensor_list), max_size
    tensor = torch.zeros([b, c, h, w], dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=tensor_list[0].device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)


def _bbox_pairs(boxes, query_boxes, mode):
    a = torch.as_tensor(boxes).to(torch.float64)
    q = torch.as_tensor(query_boxes, device=a.device).to(torch.float64)
    if a.is_cuda:
        # device routine (csrc/postprocess.hip), bit-identical to the reference's Cython loops
        from . import _lib
        a, q = a.contiguous(), q.contiguous()
        out = torch.empty(a.shape[0], q.shape[0], dtype=torch.float64, device=a.device)
        st = _lib.lib().egtr_bbox_overlaps_f64(torch.cuda.current_stream().cuda_stream, a.data_ptr(), q.data_ptr(),
                                               a.shape[0], q.shape[0], mode, out.data_ptr())
        _lib.check(st, "egtr_bbox_overlaps_f64")
        return out
    # host tensors (the reference's own habitat for this routine): the same arithmetic, vectorised
    iw = torch.minimum(a[:, None, 2], q[None, :, 2]) - torch.maximum(a[:, None, 0], q[None, :, 0]) + 1
    ih = torch.minimum(a[:, None, 3], q[None, :, 3]) - torch.maximum(a[:, None, 1], q[None, :, 1]) + 1
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_q = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    inter = iw * ih
    den = (area_a[:, None] + area_q[None, :] - inter) if mode == 0 else area_q[None, :].expand_as(inter)
    return torch.where((iw > 0) & (ih > 0), inter / den, torch.zeros((), dtype=torch.float64, device=a.device))


def bbox_overlaps(boxes, query_boxes):
    """IoU matrix [N, K] with the "+1 pixel" box convention of the reference's Cython routine
    (lib/fpn/box_intersections_cpu/bbox.pyx:21-61, used by lib/evaluation/sg_eval.py:318-322); zero where the boxes do
    not overlap; float64 like the reference (``np.float``).  Boxes on the GPU run the HIP routine
    ``egtr_bbox_overlaps_f64``; host tensors the vectorised host arithmetic."""
    return _bbox_pairs(boxes, query_boxes, 0)


def bbox_intersections(boxes, query_boxes):
    """Fraction of each query box covered by each box (bbox.pyx:64-108), same conventions as ``bbox_overlaps``."""
    return _bbox_pairs(boxes, query_boxes, 1)
