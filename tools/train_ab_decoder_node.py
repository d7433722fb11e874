#!/usr/bin/env python3
"""A/B of the decoder-layer training node on one box: bench.py --mode train with egtr_amd.ops.DECODER_TRAIN_FUSED on / off
(a module attribute, not an environment switch), alternating, two rounds.   python tools/train_ab_decoder_node.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys; sys.path.insert(0, %r); import egtr_amd.ops as o; o.DECODER_TRAIN_FUSED = %s; import bench; "
        "sys.argv = ['bench.py', '--mode', 'train', '--no-cpu-baseline', '--no-kernel-probes', '--steps', '30']; bench.main()")
for rnd in range(2):
    for on in (True, False):
        r = subprocess.run([sys.executable, "-c", CODE % (ROOT, on)], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", on, r.stderr[-800:])
            continue
        d = json.loads(line[-1])
        print(f"round {rnd} decoder node {'ON ' if on else 'OFF'}: {d['ms_per_step']:.3f} ms per step, {d['value']:.2f} images/s")
