#!/usr/bin/env python3
"""Eager vs HIP-graph replay of the small model of tests/test_gpu_model.py::test_graph_replay_matches_eager, many times:
prints the largest difference seen and how often it exceeded 1e-5 (debugging aid for an intermittent test failure)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import helpers as Hh  # noqa: E402
import weights as W  # noqa: E402
from egtr_amd.egtr import DetrForSceneGraphGeneration  # noqa: E402
from egtr_amd.runtime import GraphedForward  # noqa: E402

DEV = "cuda:0"
cfg_dict = dict(num_queries=40, encoder_layers=1, decoder_layers=2, dropout=0.0, auxiliary_loss=False, num_labels=20,
                num_rel_labels=9, use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False,
                logit_adj_tau=0.3)
torch.manual_seed(0)
model = DetrForSceneGraphGeneration(Hh.product_config(cfg_dict), fg_matrix=W.fg_matrix(20, 9)).to(DEV).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
worst, bad, worst_ee, worst_dec = 0.0, 0, 0.0, 0.0
g = GraphedForward(model, enabled=os.environ.get("NO_GRAPH") != "1")
from egtr_amd import decoder_fused
for it in range(n):
    torch.manual_seed(it)
    pv = torch.randn(1, 3, 160, 224, device=DEV)
    pm = torch.ones(1, 160, 224, dtype=torch.long, device=DEV)
    with torch.no_grad():
        e = model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
        e2 = model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
        base = model.model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
        enc = (base.encoder_last_hidden_state,)
        d1 = model.model(pixel_values=pv, pixel_mask=pm, encoder_outputs=enc, output_attention_states=True)
        d2 = model.model(pixel_values=pv, pixel_mask=pm, encoder_outputs=enc, output_attention_states=True)
    r = g(pv, pm)
    d = float((r.pred_rel - e.pred_rel).abs().max())
    worst = max(worst, d)
    bad += d >= 1e-5
    worst_ee = max(worst_ee, float((e2.pred_rel - e.pred_rel).abs().max()))
    worst_dec = max(worst_dec, float((d1.last_hidden_state - d2.last_hidden_state).abs().max()))
print("eager workspace status", decoder_fused.read_status(torch.device(DEV)), "pool left",
      {k: len(v) for k, v in decoder_fused._POOL.items()})
print(f"{n} inputs: graph vs eager max {worst:.3e} ({bad} above 1e-5); eager vs eager max {worst_ee:.3e}; decoder on a fixed "
      f"encoder output, run to run: {worst_dec:.3e}")
