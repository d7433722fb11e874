#!/usr/bin/env python3
"""Full base-model forwards twice per input: where do two runs part (debugging aid)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import helpers as Hh  # noqa: E402
import weights as W  # noqa: E402
from egtr_amd.egtr import DetrForSceneGraphGeneration  # noqa: E402
from egtr_amd import decoder_fused  # noqa: E402

DEV = "cuda:0"
cfg_dict = dict(num_queries=40, encoder_layers=1, decoder_layers=2, dropout=0.0, auxiliary_loss=False, num_labels=20,
                num_rel_labels=9, use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False,
                logit_adj_tau=0.3)
torch.manual_seed(0)
model = DetrForSceneGraphGeneration(Hh.product_config(cfg_dict), fg_matrix=W.fg_matrix(20, 9)).to(DEV).eval()
seen = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    torch.manual_seed(it)
    pv = torch.randn(1, 3, 160, 224, device=DEV)
    pm = torch.ones(1, 160, 224, dtype=torch.long, device=DEV)
    with torch.no_grad():
        a = model.model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
        b = model.model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
    def diffs(x, y):
        out = {"enc": float((x.encoder_last_hidden_state - y.encoder_last_hidden_state).abs().max())}
        for i, (s, t) in enumerate(zip(x.decoder_hidden_states, y.decoder_hidden_states)):
            out[f"state{i}"] = float((s - t).abs().max())
        for i, (s, t) in enumerate(zip(x.decoder_attention_queries, y.decoder_attention_queries)):
            out[f"q{i}"] = float((s - t).abs().max())
        return out
    d = diffs(a, b)
    if max(v for k, v in d.items() if k != "enc") > 1e-4:
        bad = a if not torch.isfinite(b.last_hidden_state).all() else b
        print(it, {k: f"{v:.2e}" for k, v in d.items()}, "finite", bool(torch.isfinite(a.last_hidden_state).all()),
              bool(torch.isfinite(b.last_hidden_state).all()), "status", decoder_fused.read_status(torch.device(DEV)))
        s1, t1 = a.decoder_hidden_states[1], b.decoder_hidden_states[1]
        rows = (s1 - t1).abs().amax(-1).flatten().nonzero().flatten().tolist()
        print("   rows differing in state1:", rows[:40], "cols:", (s1 - t1).abs().amax((0, 1)).nonzero().flatten().tolist()[:20])
print("done")
