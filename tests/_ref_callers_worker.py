"""Child process of tests/test_reference_callers_cpu.py (a fresh interpreter: the reference's top-level package names --
model, data, lib, util -- must not leak into the test session).

Imports the REFERENCE's own drivers, /root/reference/train_egtr.py and evaluate_egtr.py, with THIS repository first on
sys.path, so that their ``from model.deformable_detr import ...`` / ``from model.egtr import ...`` (train_egtr.py:31-36)
resolve to the product (model/ -> egtr_amd), and runs the reference's code against it:
  SGG.__init__ (train_egtr.py:189-278: config plumbing, from_pretrained with ignore_mismatched_sizes / loading info),
  SGG.configure_optimizers (:426-467), SGG.common_step (:303-319), evaluate_egtr.calculate_fps (:26-36).
Nothing of the reference is copied: what this box lacks (pytorch-lightning, torchvision, pycocotools, the dataset / evaluator
modules -- none of them on these code paths) is replaced by inert stand-ins for the import, the three HIP ops by the
oracle-built CPU stand-ins of tests/cpu_kernels.py.  Prints one JSON line."""
import json
import os
import sys
import tempfile
import types
from unittest import mock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [ROOT, HERE, os.path.join(HERE, "golden")]
sys.path.append(REF)   # AFTER the repository: `model` is the product's alias package, data / lib / util are the reference's

import _ref_import  # noqa: E402  (only its stub-backbone class and the shims of removed transformers names are used)
import helpers as Hh  # noqa: E402
import weights as W  # noqa: E402


def install_stubs():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):   # what SGG uses of it: nn.Module behaviour, log / log_dict, global_step
        global_step = 0

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Trainer = mock.MagicMock(name="Trainer")
    pl.seed_everything = lambda *a, **k: None
    sys.modules["pytorch_lightning"] = pl
    for name in ("pytorch_lightning.callbacks", "pytorch_lightning.callbacks.early_stopping", "pytorch_lightning.loggers",
                 "pytorch_lightning.strategies", "pytorch_lightning.strategies.ddp", "pytorch_lightning.utilities",
                 "pytorch_lightning.utilities.rank_zero", "torchvision", "torchvision.ops", "torchvision.ops.boxes",
                 "data.open_image", "data.visual_genome", "lib.evaluation.coco_eval", "lib.evaluation.oi_eval",
                 "lib.evaluation.sg_eval", "util.misc"):
        sys.modules[name] = mock.MagicMock(name=name)
    sys.modules["pytorch_lightning.utilities.rank_zero"].rank_zero_only = lambda f: f


def main():
    install_stubs()
    import egtr_amd.deformable_detr as pdd
    import egtr_amd.ops as ops
    import cpu_kernels as ck
    ops._msda = lambda: ck.OracleMSDA
    ops.decoder_self_attention = ck.decoder_self_attention
    ops.relation_head = ck.relation_head
    import train_egtr
    import evaluate_egtr
    import model.egtr as alias
    res = {"train_egtr": os.path.abspath(train_egtr.__file__), "model_pkg": os.path.abspath(alias.__file__),
           "sgg_class_is_real": isinstance(train_egtr.SGG, type)}

    g = Hh.load_golden(os.path.join(HERE, "golden"), "sgg_small.npz")
    cfg_dict, shapes = json.loads(str(g["cfg"])), json.loads(str(g["shapes"]))
    pdd.DeformableDetrTimmConvEncoder = _ref_import.make_stub_backbone_class()   # the fixtures' backbone (timm is absent)
    with tempfile.TemporaryDirectory() as tmp:
        # the "pretrained Deformable DETR" SGG starts from (pretrain_detr.py's product): another label count, no relation head
        cfg0 = Hh.product_config({**cfg_dict, "num_labels": 5})
        torch.manual_seed(0)
        pdd.DeformableDetrForObjectDetection(cfg0).save_pretrained(tmp)
        fg = W.fg_matrix(cfg_dict["num_labels"], cfg_dict["num_rel_labels"], seed=0)
        sgg = train_egtr.SGG(
            architecture="SenseTime/deformable-detr", backbone_dirpath=None, auxiliary_loss=False, lr=2e-6, lr_backbone=2e-7,
            lr_initialized=2e-4, weight_decay=1e-4, pretrained=tmp, main_trained="", from_scratch=False,
            id2label={i: str(i) for i in range(cfg_dict["num_labels"])}, rel_loss_coefficient=15.0, smoothing=1e-14,
            rel_sample_negatives=80, rel_sample_nonmatching=80, rel_categories=[str(i) for i in range(cfg_dict["num_rel_labels"])],
            multiple_sgg_evaluator=None, multiple_sgg_evaluator_list=[], single_sgg_evaluator=None, single_sgg_evaluator_list=[],
            coco_evaluator=None, oi_evaluator=None, feature_extractor=None, num_queries=cfg_dict["num_queries"],
            ce_loss_coefficient=2.0, rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True, use_freq_bias=True,
            fg_matrix=fg, use_log_softmax=False, freq_bias_eps=1e-12, connectivity_loss_coefficient=30.0,
            logit_adjustment=False, logit_adj_tau=0.3)
    res["model_class"] = f"{type(sgg.model).__module__}.{type(sgg.model).__name__}"
    res["initialized_keys"] = sorted(sgg.initialized_keys)
    res["config"] = {k: getattr(sgg.config, k) for k in ("num_labels", "num_rel_labels", "num_queries", "auxiliary_loss",
                                                            "rel_loss_coefficient", "connectivity_loss_coefficient")}

    # ---- configure_optimizers (train_egtr.py:426-467) vs the product trainer's groups on the same module
    opt = sgg.configure_optimizers()
    names = {id(p): n for n, p in sgg.named_parameters()}
    res["groups"] = [{"lr": gr["lr"], "weight_decay": gr["weight_decay"], "n": len(gr["params"]),
                      "names": sorted(names[id(p)] for p in gr["params"])} for gr in opt.param_groups]
    from egtr_amd import runtime
    own = runtime.configure_optimizers(sgg, lr=2e-6, lr_backbone=2e-7, lr_initialized=2e-4, weight_decay=1e-4,
                                       initialized_keys=sgg.initialized_keys)
    own = own[0] if isinstance(own, tuple) else own
    res["own_groups"] = [{"lr": gr["lr"], "n": len(gr["params"]), "names": sorted(names[id(p)] for p in gr["params"])}
                         for gr in own.param_groups]

    # ---- common_step (train_egtr.py:303-319) on the fixture's weights, inputs and targets
    sd = W.fill_state_dict(shapes, seed=int(g["seed"]), alias_heads=True)
    sd["triplet_dist"], sd["rel_dist"] = W.freq_bias_tables(fg, cfg_dict["freq_bias_eps"])
    sgg.model.load_state_dict(sd)
    pv, pm = Hh.small_inputs(g)
    targets = W.make_targets(int(g["target_seed"]), 2, cfg_dict["num_queries"], cfg_dict["num_labels"],
                             cfg_dict["num_rel_labels"])
    batch = {"pixel_values": pv, "pixel_mask": pm, "labels": targets}
    for training in (False, True):
        sgg.train(training)
        with torch.set_grad_enabled(training):
            loss, loss_dict = sgg.common_step(batch, 0)
        key = "train" if training else "eval"
        res[f"{key}_loss"] = float(loss)
        res[f"{key}_loss_dict"] = {k: float(v) for k, v in loss_dict.items()}
    # training_step exists and logs through the LightningModule interface
    res["training_step_loss"] = float(sgg.training_step(batch, 0))

    # ---- calculate_fps (evaluate_egtr.py:26-36): the reference moves the batch with .cuda(); on this GPU-less box the call is
    # a no-op so that the same lines run on the CPU stand-ins
    calls = []
    orig_forward = sgg.model.forward

    def counting(*a, **k):
        calls.append(sorted(k))
        return orig_forward(*a, **k)

    sgg.model.forward = counting
    with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        evaluate_egtr.calculate_fps(sgg.model, [batch, batch])
    res["fps_calls"] = calls
    res["fps_eval_mode"] = not sgg.model.training
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
