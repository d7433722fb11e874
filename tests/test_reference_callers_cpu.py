"""CPU, build container only: the REFERENCE's own drivers run against the product package.

north_star: "keeping the DeformableDetr / SceneGraphGeneration module API so it drops into train_egtr.py and evaluate_egtr.py
unchanged".  tests/_ref_callers_worker.py imports /root/reference/train_egtr.py and evaluate_egtr.py in a fresh interpreter
with this repository first on sys.path (their ``from model... import`` lines then bind to model/ -> egtr_amd) and runs
SGG.__init__ (train_egtr.py:189-278), configure_optimizers (:426-467), common_step (:303-319) and calculate_fps
(evaluate_egtr.py:26-36).  Skipped where /root/reference is absent (the GPU box); nothing of the reference is stored here."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only in the build container")


@pytest.fixture(scope="module")
def run():
    env = dict(os.environ, PYTHONPATH="")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_ref_callers_worker.py")], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_reference_drivers_bind_to_the_product_package(run):
    assert run["train_egtr"] == "/root/reference/train_egtr.py"
    assert run["model_pkg"] == os.path.join(ROOT, "model", "egtr.py")
    assert run["sgg_class_is_real"]
    assert run["model_class"] == "egtr_amd.egtr.DetrForSceneGraphGeneration"


def test_sgg_constructor_config_plumbing_and_loading_info(run, golden_dir):
    g = np.load(os.path.join(golden_dir, "sgg_small.npz"), allow_pickle=False)
    cfg = json.loads(str(g["cfg"]))
    for k in ("num_labels", "num_rel_labels", "num_queries", "auxiliary_loss", "rel_loss_coefficient",
              "connectivity_loss_coefficient"):
        assert run["config"][k] == cfg[k], k
    init = set(run["initialized_keys"])   # missing + size-mismatched keys of the 5-class detection checkpoint
    assert {"class_embed.0.weight", "class_embed.0.bias", "rel_dist", "triplet_dist"} <= init
    assert any(k.startswith("rel_predictor") or k.startswith("connectivity_layer") for k in init)
    assert not any(k.startswith("model.backbone") or k.startswith("model.encoder") for k in init)


def test_configure_optimizers_groups_equal_the_product_trainers(run):
    ref, own = run["groups"], run["own_groups"]
    assert len(ref) == len(own) == 3
    assert [g["lr"] for g in ref] == [2e-6, 2e-7, 2e-4] and all(g["weight_decay"] == 1e-4 for g in ref)
    for a, b in zip(ref, own):
        assert a["lr"] == b["lr"] and a["names"] == b["names"]
    assert all("backbone" in n or "reference_points" in n or "sampling_offsets" in n for n in ref[1]["names"])
    assert sum(g["n"] for g in ref) == len({n for g in ref for n in g["names"]})   # every parameter in exactly one group


def test_common_step_losses_equal_the_reference_fixture(run, golden_dir):
    g = np.load(os.path.join(golden_dir, "sgg_small.npz"), allow_pickle=False)
    for key in ("eval", "train"):
        ref = json.loads(str(g[f"{key}_loss_dict"]))
        got = run[f"{key}_loss_dict"]
        assert set(ref) == set(got)
        for k, v in ref.items():
            assert abs(got[k] - v) < 3e-4 * max(1.0, abs(v)), (key, k, got[k], v)
        assert abs(run[f"{key}_loss"] - float(g[f"{key}_loss"])) < 3e-4 * abs(float(g[f"{key}_loss"]))
    assert abs(run["training_step_loss"] - run["train_loss"]) < 1e-6 * abs(run["train_loss"])


def test_calculate_fps_runs_the_product_forward(run):
    assert run["fps_eval_mode"]
    want = sorted(["pixel_values", "pixel_mask", "output_attentions", "output_attention_states", "output_hidden_states"])
    assert run["fps_calls"] == [want, want]
