"""Golden vectors for the OTHER two branches of the reference's evaluate_batch (train_egtr.py:120-139 single-predicate
evaluator, :154-174 Open Images evaluator), produced by RUNNING THE REFERENCE exactly as make_golden_post.py does for the
multiple-predicate branch (same inputs: weights.post_inputs(61); same inert mocks for the Lightning / torchvision imports;
recording stand-ins for the evaluators).

    python tests/golden/make_golden_post_branches.py      -> tests/golden/postprocess_branches.npz
The OI entry's [N*N, R] score matrix is stored as a strided sample + checksums (the full matrix is pred_rel itself)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_post import Recorder, import_train_egtr  # noqa: E402
import weights as W  # noqa: E402


class OIRecorder:
    def __init__(self):
        self.entries = []

    def __call__(self, gt_entry, pred_entry):
        self.entries.append((gt_entry, pred_entry))


def main():
    te = import_train_egtr()
    outputs, targets, meta = W.post_inputs(seed=61)
    single, oi = Recorder(), OIRecorder()
    te.evaluate_batch(outputs, targets, None, [], {"sgdet": single}, [], oi, meta["num_labels"], max_topk=100)
    assert len(single.entries) == len(targets) == len(oi.entries)
    res = {}
    for j, (_, pred) in enumerate(single.entries):
        for k, v in pred.items():
            res[f"single{j}_{k}"] = np.asarray(v)
    for j, (_, pred) in enumerate(oi.entries):
        ps = np.asarray(pred["pred_scores"])
        inds = np.asarray(pred["sbj_obj_inds"])
        res[f"oi{j}_pred_scores_strided"] = ps[::97]
        res[f"oi{j}_pred_scores_sum"] = np.float64(ps.astype(np.float64).sum())
        res[f"oi{j}_pred_scores_shape"] = np.asarray(ps.shape)
        res[f"oi{j}_sbj_obj_inds_strided"] = inds[::97]
        res[f"oi{j}_sbj_obj_inds_checksum"] = np.int64((inds.astype(np.int64) * np.array([1000003, 7])).sum())
        res[f"oi{j}_pred_classes"] = np.asarray(pred["pred_classes"])
        res[f"oi{j}_obj_scores"] = np.asarray(pred["obj_scores"])
        res[f"oi{j}_pred_boxes"] = np.asarray(pred["pred_boxes"])
    np.savez_compressed(os.path.join(HERE, "postprocess_branches.npz"), seed=61, **res)
    print({k: getattr(v, "shape", v) for k, v in res.items()})


if __name__ == "__main__":
    main()
