#!/usr/bin/env python3
"""Micro-benchmark of the fused relation-head kernel at the VG shape (N=200, T=7, R=50)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))


def main():
    import test_gpu_kernels as T
    from egtr_amd.ops import relation_head
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    d, trip, node = T._head_inputs(60, B, 200, 7, 50, 150)
    dd = {k: v.cuda() for k, v in d.items()}
    trip, node = trip.cuda(), node.cuda()
    for _ in range(5):
        relation_head(*dd.values(), trip, node, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        relation_head(*dd.values(), trip, node, False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * B * 200 * 200 * (2 * 256 * 256 + 256 * 64)
    print(f"rel_head fwd B={B}: {us:.1f} us/launch, {fl / us / 1e6:.1f} TFLOP/s (f32 MFMA peak 157)")
    from egtr_amd import ops
    w2xr, w3xr, w2xc = ops.rel_head_split_weights(dd["w2r"], dd["w3r"], dd["w2c"])
    args = (dd["gate_q"], dd["gate_k"], dd["uq"], dd["uk"], dd["b1"], w2xr, dd["b2r"], w3xr, dd["b3r"], w2xc, dd["b2c"],
            dd["w3c"], dd["b3c"], 50, trip, node, False)
    for _ in range(5):
        ops.relation_head_split_bf16(*args)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        ops.relation_head_split_bf16(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"rel_head fwd B={B} split-bf16 (x6): {us:.1f} us/launch, {fl / us / 1e6:.1f} TFLOP/s algorithmic")
    from egtr_amd.ops import relation_head_bf16w
    wn = ("w2r", "w3r", "w2c", "w3c")
    db = {k: (v.bfloat16() if k in wn else v) for k, v in dd.items()}
    for _ in range(5):
        relation_head_bf16w(*db.values(), trip, node, False)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        relation_head_bf16w(*db.values(), trip, node, False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"rel_head fwd bf16 matrix cores B={B}: {us:.1f} us/launch, {fl / us / 1e6:.1f} TFLOP/s")
    # training: forward (saving h1 / h2) + backward (rocBLAS GEMMs + the HIP pairwise kernels) vs autograd through the
    # PyTorch statement of the same separable algebra (what round 1 shipped first)
    dg = {k: v.clone().requires_grad_(True) for k, v in dd.items()}

    def step_hip():
        rel, conn, _ = relation_head(*dg.values(), trip, node, False)
        (rel.sum() + conn.sum()).backward()

    def separable(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c):
        g = torch.sigmoid(gate_q[:, :, None, :] + gate_k[:, None, :, :])
        h1 = torch.relu(torch.einsum("bijt,bitc->bijc", g, uq) + torch.einsum("bijt,bjtc->bijc", g, uk) + b1)
        lin = torch.nn.functional.linear
        rel = lin(torch.relu(lin(h1[..., :256], w2r, b2r)), w3r, b3r)
        conn = lin(torch.relu(lin(h1[..., 256:], w2c, b2c)), w3c, b3c)
        return rel, conn

    def step_torch():
        rel, conn = separable(*dg.values())
        (rel.sum() + conn.sum()).backward()

    for name, fn in (("HIP fwd+bwd", step_hip), ("PyTorch separable fwd+bwd", step_torch)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"rel_head {name} B={B}: {e0.elapsed_time(e1) / 10:.3f} ms/step")


if __name__ == "__main__":
    main()
