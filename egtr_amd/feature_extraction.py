"""Tensor-based counterparts of the reference's feature extractors (model/deformable_detr.py:270-385).

The reference subclasses transformers-4.18 ``DetrFeatureExtractor`` (PIL + torchvision pipeline).  Neither PIL
augmentation nor torchvision is part of the hot path; what the drivers need from these classes is
(1) resize (shorter side ``size``, longer side capped at ``max_size``) + ImageNet normalisation,
(2) ``pad_and_create_pixel_mask`` (collate_fn, train_egtr.py:176-186) and (3) ``post_process`` (dd:273-312).
Those are provided on torch tensors.  The ``WithAugmentor`` variants add the random horizontal flip / random
resize of dd:319-385 in tensor form (crop only for the non-"NoCrop" class).
"""
import random

import torch
import torch.nn.functional as F

from .util import center_to_corners_format

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _target_size(h, w, size, max_size):
    """DETR resize rule (shorter side -> size, longer side <= max_size)."""
    mn, mx = float(min(h, w)), float(max(h, w))
    if max_size is not None and mx / mn * size > max_size:
        size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


class DeformableDetrFeatureExtractor:
    model_input_names = ["pixel_values", "pixel_mask"]

    def __init__(self, size=800, max_size=1333, do_resize=True, do_normalize=True, image_mean=IMAGENET_MEAN,
                 image_std=IMAGENET_STD, format="coco_detection", **kwargs):
        self.size, self.max_size = size, max_size
        self.do_resize, self.do_normalize = do_resize, do_normalize
        self.image_mean, self.image_std = tuple(image_mean), tuple(image_std)
        self.format = format

    @classmethod
    def from_pretrained(cls, name_or_path=None, **kwargs):
        return cls(**kwargs)

    # ---- geometry helpers on (C,H,W) float tensors in [0,1]; target boxes are absolute xyxy until normalised
    def _resize(self, image, target, size, max_size=None):
        h, w = image.shape[-2:]
        nh, nw = _target_size(h, w, size, max_size)
        image = F.interpolate(image[None], size=(nh, nw), mode="bilinear", align_corners=False)[0]
        if target is not None:
            target = dict(target)
            if "boxes" in target:
                target["boxes"] = target["boxes"] * torch.tensor([nw / w, nh / h, nw / w, nh / h])
            target["size"] = torch.tensor([nh, nw])
        return image, target

    def _normalize(self, image, target):
        mean = torch.tensor(self.image_mean).view(-1, 1, 1)
        std = torch.tensor(self.image_std).view(-1, 1, 1)
        image = (image - mean) / std
        if target is not None and "boxes" in target:
            h, w = image.shape[-2:]
            b = target["boxes"]
            cxcywh = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0],
                                  b[:, 3] - b[:, 1]], -1)
            target = dict(target)
            target["boxes"] = cxcywh / torch.tensor([w, h, w, h], dtype=torch.float32)
        return image, target

    def _augment(self, image, target):
        return image, target

    def __call__(self, images, annotations=None, return_tensors="pt", **kwargs):
        single = torch.is_tensor(images) and images.dim() == 3
        images = [images] if single else list(images)
        annotations = [annotations] if (single and annotations is not None) else annotations
        out_images, out_targets = [], []
        for i, img in enumerate(images):
            img = torch.as_tensor(img, dtype=torch.float32)
            tgt = annotations[i] if annotations is not None else None
            if tgt is not None and "orig_size" not in tgt:
                tgt = dict(tgt, orig_size=torch.tensor(img.shape[-2:]))
            img, tgt = self._augment(img, tgt)
            if self.do_resize:
                img, tgt = self._resize(img, tgt, self.size, self.max_size)
            if self.do_normalize:
                img, tgt = self._normalize(img, tgt)
            out_images.append(img)
            out_targets.append(tgt)
        enc = self.pad_and_create_pixel_mask(out_images)
        if annotations is not None:
            enc["labels"] = out_targets
        return enc

    def pad_and_create_pixel_mask(self, pixel_values_list, return_tensors="pt"):
        """Pad to the largest H, W in the batch (top-left aligned); mask 1 = real pixel, 0 = padding.  Images that already
        live on the GPU (fp32 [C, h, w]) are batched there by one HIP launch (egtr_pad_batch_f32); host images take the
        host loop, like the reference."""
        mh = max(int(x.shape[-2]) for x in pixel_values_list)
        mw = max(int(x.shape[-1]) for x in pixel_values_list)
        b = len(pixel_values_list)
        c = pixel_values_list[0].shape[0]
        if all(torch.is_tensor(x) and x.is_cuda for x in pixel_values_list):
            from . import _lib
            from .load_custom import _stream
            dev = pixel_values_list[0].device
            imgs = [x.to(dtype=torch.float32).contiguous() for x in pixel_values_list]
            ptrs = torch.tensor([x.data_ptr() for x in imgs], dtype=torch.int64).to(dev, non_blocking=True)
            hw = torch.tensor([[int(x.shape[-2]), int(x.shape[-1])] for x in imgs], dtype=torch.int32).to(dev, non_blocking=True)
            pv = torch.empty(b, c, mh, mw, dtype=torch.float32, device=dev)
            pm = torch.empty(b, mh, mw, dtype=torch.int64, device=dev)
            _lib.check(_lib.lib().egtr_pad_batch_f32(_stream(), ptrs.data_ptr(), hw.data_ptr(), b, c, mh, mw,
                                                     pv.data_ptr(), pm.data_ptr()), "egtr_pad_batch_f32")
            del imgs   # (alive until the launch was enqueued on the stream that also frees them)
            return {"pixel_values": pv, "pixel_mask": pm}
        pv = torch.zeros(b, c, mh, mw, dtype=torch.float32)
        pm = torch.zeros(b, mh, mw, dtype=torch.int64)
        for i, x in enumerate(pixel_values_list):
            h, w = x.shape[-2:]
            pv[i, :, :h, :w] = torch.as_tensor(x, dtype=torch.float32)
            pm[i, :h, :w] = 1
        return {"pixel_values": pv, "pixel_mask": pm}

    def post_process(self, outputs, target_sizes):
        """dd:273-312: top-100 (query, class) pairs by sigmoid score, boxes to absolute xyxy."""
        out_logits, out_bbox = outputs.logits, outputs.pred_boxes
        if len(out_logits) != len(target_sizes):
            raise ValueError("Make sure that you pass in as many target sizes as the batch dimension of the logits")
        if target_sizes.shape[1] != 2:
            raise ValueError("Each element of target_sizes must contain the size (h, w) of each image of the batch")
        prob = out_logits.sigmoid()
        topk_values, topk_indexes = torch.topk(prob.view(out_logits.shape[0], -1), 100, dim=1)
        scores = topk_values
        topk_boxes = torch.div(topk_indexes, out_logits.shape[2], rounding_mode="floor")
        labels = topk_indexes % out_logits.shape[2]
        boxes = center_to_corners_format(out_bbox)
        boxes = torch.gather(boxes, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.unbind(1)
        scale_fct = torch.stack([img_w, img_h, img_w, img_h], dim=1).to(boxes.device)
        boxes = boxes * scale_fct[:, None, :]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]


class DeformableDetrFeatureExtractorWithAugmentorNoCrop(DeformableDetrFeatureExtractor):
    """Random horizontal flip + random shorter-side scale (dd:352-385), tensor form."""
    scales = [480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800]
    use_crop = False

    def _hflip(self, image, target):
        image = image.flip(-1)
        if target is not None and "boxes" in target:
            w = image.shape[-1]
            b = target["boxes"]
            target = dict(target, boxes=torch.stack([w - b[:, 2], b[:, 1], w - b[:, 0], b[:, 3]], -1))
        return image, target

    def _augment(self, image, target):
        if random.random() < 0.5:
            image, target = self._hflip(image, target)
        if self.use_crop and random.random() < 0.5:
            image, target = self._resize(image, target, random.choice([400, 500, 600]))
            image, target = self._random_crop(image, target, 384, 600)
        return image, target

    def _resize(self, image, target, size, max_size=None):
        if size == self.size:  # the final resize of the pipeline draws a random scale (dd:340,378)
            size, max_size = random.choice(self.scales), 1333
        return super()._resize(image, target, size, max_size)

    def _random_crop(self, image, target, min_size, max_size):
        h, w = image.shape[-2:]
        cw = random.randint(min_size, min(w, max_size))
        ch = random.randint(min_size, min(h, max_size))
        top, left = random.randint(0, h - ch), random.randint(0, w - cw)
        image = image[:, top:top + ch, left:left + cw]
        if target is not None and "boxes" in target:
            b = target["boxes"] - torch.tensor([left, top, left, top], dtype=torch.float32)
            b = torch.min(b.reshape(-1, 2, 2), torch.tensor([cw, ch], dtype=torch.float32)).clamp(min=0).reshape(-1, 4)
            keep = (b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])
            target = dict(target, boxes=b[keep], size=torch.tensor([ch, cw]))
            for f in ("class_labels", "area", "iscrowd"):
                if f in target:
                    target[f] = target[f][keep]
        return image, target


class DeformableDetrFeatureExtractorWithAugmentor(DeformableDetrFeatureExtractorWithAugmentorNoCrop):
    use_crop = True
