#!/usr/bin/env python3
"""Which output of the cluster decoder differs between two runs on the same encoder output (debugging aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import helpers as Hh  # noqa: E402
import weights as W  # noqa: E402
from egtr_amd.egtr import DetrForSceneGraphGeneration  # noqa: E402

DEV = "cuda:0"
nq, nl = int(sys.argv[1]), int(sys.argv[2])
cfg_dict = dict(num_queries=nq, encoder_layers=1, decoder_layers=nl, dropout=0.0, auxiliary_loss=False, num_labels=20,
                num_rel_labels=9, use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False,
                logit_adj_tau=0.3)
torch.manual_seed(0)
model = DetrForSceneGraphGeneration(Hh.product_config(cfg_dict), fg_matrix=W.fg_matrix(20, 9)).to(DEV).eval()
torch.manual_seed(1)
pv = torch.randn(1, 3, 160, 224, device=DEV)
pm = torch.ones(1, 160, 224, dtype=torch.long, device=DEV)
with torch.no_grad():
    base = model.model(pixel_values=pv, pixel_mask=pm, output_attention_states=True, output_hidden_states=True)
    enc = (base.encoder_last_hidden_state,)
    ref = None
    counts = {}
    for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 200):
        o = model.model(pixel_values=pv, pixel_mask=pm, encoder_outputs=enc, output_attention_states=True,
                        output_hidden_states=True)
        cur = {f"state{i}": t.clone() for i, t in enumerate(o.decoder_hidden_states)}
        cur.update({f"q{i}": t.clone() for i, t in enumerate(o.decoder_attention_queries)})
        cur.update({f"k{i}": t.clone() for i, t in enumerate(o.decoder_attention_keys)})
        if ref is None:
            ref = cur
            continue
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                d = (ref[k] - cur[k]).abs()
                rows = d.reshape(-1, d.shape[-1]).amax(-1).nonzero().flatten().tolist() if k.startswith("state") else []
                counts.setdefault(k, []).append((it, float(d.max()), rows[:12]))
for k, v in counts.items():
    print(k, len(v), v[:3])
print("differing tensors:", sorted(counts))
