"""Golden vectors for the post-processing rows, produced by RUNNING THE REFERENCE:

  * train_egtr.evaluate_batch (train_egtr.py:43-106, multiple-predicate branch) is imported from /root/reference and
    called with a recording stand-in for the evaluator, so the fixture holds exactly the ``pred_entry`` the reference
    hands to ``BasicSceneGraphEvaluator.evaluate_scene_graph_entry`` (top-100 triplets via lib.pytorch_misc.argsort_desc,
    rel scores, rescaled boxes, classes, object scores).  train_egtr.py imports pytorch-lightning / torchvision /
    pycocotools-based modules at import time, none of which is on the evaluate_batch path: they are replaced by inert
    mock modules for the import (no reference code is copied or re-implemented here).
  * bbox_overlaps / bbox_intersections come from the reference's Cython source compiled by oracle/Makefile
    (oracle/_ref/bbox*.so).

    make -C oracle ref && python tests/golden/make_golden_post.py      -> tests/golden/postprocess.npz
Inputs are regenerated from seeds by the tests (weights.post_inputs), so the fixture stores outputs only."""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402
import weights as W  # noqa: E402


def import_train_egtr():
    _ref_import.load_reference()  # model.* under the transformers shims, sys.path -> /root/reference
    for name in ("pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.early_stopping",
                 "pytorch_lightning.loggers", "pytorch_lightning.strategies", "pytorch_lightning.strategies.ddp",
                 "pytorch_lightning.utilities", "pytorch_lightning.utilities.rank_zero", "torchvision",
                 "torchvision.ops", "torchvision.ops.boxes", "data.open_image", "data.visual_genome",
                 "lib.evaluation.coco_eval", "lib.evaluation.oi_eval", "lib.evaluation.sg_eval", "util.misc"):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    import train_egtr
    return train_egtr


class Recorder:
    def __init__(self):
        self.entries = []

    def evaluate_scene_graph_entry(self, gt_entry, pred_entry):
        self.entries.append((gt_entry, pred_entry))


def main():
    te = import_train_egtr()
    from oracle import ref_bbox
    bbox = ref_bbox.load()
    assert bbox is not None, "run `make -C oracle ref` first"
    res = {}
    # ---- evaluate_batch: two images, VG-sized heads, incl. exact score ties (quantised relation scores in image 1)
    outputs, targets, meta = W.post_inputs(seed=61)
    rec = Recorder()
    te.evaluate_batch(outputs, targets, {"sgdet": rec}, [], None, [], None, meta["num_labels"], max_topk=100)
    assert len(rec.entries) == len(targets)
    for j, (gt, pred) in enumerate(rec.entries):
        for k, v in pred.items():
            res[f"pred{j}_{k}"] = np.asarray(v)
        for k, v in gt.items():
            res[f"gt{j}_{k}"] = np.asarray(v)
    # ---- bbox_overlaps / bbox_intersections: predicted boxes of image 0 vs its ground-truth boxes, plus edge cases
    cases = W.bbox_cases(seed=62)
    for name, (a, b) in cases.items():
        res[f"iou_{name}"] = bbox.bbox_overlaps(a, b)
        res[f"inter_{name}"] = bbox.bbox_intersections(a, b)
    a = res["pred0_pred_boxes"].astype(np.float64)
    b = res["gt0_gt_boxes"].astype(np.float64)
    res["iou_pred0_vs_gt0"] = bbox.bbox_overlaps(a, b)
    np.savez_compressed(os.path.join(HERE, "postprocess.npz"), seed=61, bbox_seed=62, **res)
    print({k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
