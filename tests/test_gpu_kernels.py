"""GPU (-m gpu): parity of every HIP kernel, called through the C ABI (ctypes -> libegtr_hip.so), against the CPU
oracle on the same seeded inputs and against the committed golden vectors from the reference.

Tolerances (fp32 path; the north-star's bar is 1e-3 on logits):
  MSDA forward 2e-5 abs on O(1) outputs; backward 2e-4 (grad_loc carries a factor W/H -> 5e-3);
  self-attention 2e-5; relation head 2e-4 on logits of magnitude O(1..10).
"""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
import hip_test_abi as TA
import weights as W
from oracle import detr as O
from oracle import msda as OM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _kernels():
    from egtr_amd.load_custom import load_hip_kernels
    return load_hip_kernels()


def _run_msda(x, bwd=True):
    k = _kernels()
    d = {n: t.to(DEV) for n, t in x.items()}
    out = k.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 64)
    res = [out.cpu()]
    if bwd:
        gv, gl, ga = k.ms_deform_attn_backward(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], d["grad_out"], 64)
        res += [gv.cpu(), gl.cpu(), ga.cpu()]
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("case", ["a", "b"])
def test_msda_vs_reference_golden(golden_dir, case):
    """Inputs regenerated from the fixture's seed; expected outputs are the REFERENCE's (fp32 run)."""
    g = Hh.load_golden(golden_dir, "msda.npz")
    c = json.loads(str(g[f"{case}_case"]))
    x = W.make_msda_inputs(c["seed"], c["B"], c["Lq"], c["M"], c["D"], [tuple(s) for s in c["shapes"]], c["P"])
    out, gv, gl, ga = _run_msda(x)
    assert (out - _t(g[f"{case}_f64_out"]).float()).abs().max() < 2e-5
    assert (gv - _t(g[f"{case}_f64_grad_value"]).float()).abs().max() < 2e-4
    assert (ga - _t(g[f"{case}_f64_grad_attn"]).float()).abs().max() < 2e-4
    assert (gl - _t(g[f"{case}_f64_grad_loc"]).float()).abs().max() < 5e-3


@pytest.mark.parametrize("B,Lq,shapes", [
    (1, 200, [(19, 32), (10, 16), (5, 8), (3, 4)]),       # decoder-like: Lq = N
    (2, 820, [(19, 32), (10, 16), (5, 8), (3, 4)]),       # encoder-like: Lq = S
    (4, 7, [(5, 5), (3, 3), (2, 2), (1, 1)]),             # ragged tail: nq not a multiple of waves/block
    (1, 1, [(1, 1), (1, 1), (1, 1), (1, 1)]),             # degenerate 1x1 levels
    (3, 33, [(8, 8), (4, 4)]),                            # L=2, P=8 (L*P = 16 fast path)
    (2, 2500, [(19, 32), (10, 16), (5, 8), (3, 4)]),      # long arbitrary query list (Lq != S, B Lq > 4096): the backward's
                                                          # one-wave-per-query form WITH its atomics (msda_bwd_q64_f32<true, 1>)
])
def test_msda_vs_oracle(B, Lq, shapes):
    P = 16 // len(shapes)
    x = W.make_msda_inputs(100 + B + Lq, B, Lq, 8, 32, shapes, P, oob_frac=0.15)
    out, gv, gl, ga = _run_msda(x)
    ref = OM.msda_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    rgv, rgl, rga = OM.msda_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], x["grad_out"])
    assert (out - ref).abs().max() < 2e-5
    assert (gv - rgv).abs().max() < 2e-4
    assert (ga - rga).abs().max() < 2e-4
    assert (gl - rgl).abs().max() < 5e-3 * max(1.0, float(rgl.abs().max()) / 50)


def _grid_inputs(seed, B, shapes, jitter, dtype=torch.float32):
    """Encoder-like operands: queries = the pixels of the levels, samples = reference + ring offsets + jitter."""
    import math
    g = torch.Generator().manual_seed(seed)
    M, L, P = 8, len(shapes), 16 // len(shapes)
    S = sum(h * w for h, w in shapes)
    refs = []
    for (h, w) in shapes:
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)
    th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    for i in range(P):
        grid[:, :, i, :] *= i + 1
    off = grid[None, None] + jitter * torch.randn(B, S, M, L, P, 2, generator=g)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    attn = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    return dict(value=torch.randn(B, S, M, 32, generator=g), shapes=shp, lsi=lsi, loc=loc.contiguous(),
                attn=attn.contiguous(), grad_out=torch.randn(B, S, 256, generator=g))


@pytest.mark.parametrize("shapes,B,jitter", [
    ([(19, 32), (10, 16), (5, 8), (3, 4)], 2, 0.3),
    ([(38, 63), (19, 32), (10, 16), (5, 8)], 1, 1.0),
    ([(75, 125), (38, 63), (19, 32), (10, 16)], 1, 0.5),   # the 600x1000 pyramid
    ([(9, 13), (5, 7)], 3, 0.5),                           # L = 2, P = 8
])
def test_msda_wave_kernel_equals_generic_kernel_on_encoder_shapes(shapes, B, jitter):
    """The two forward kernels behind the C ABI (1 = wave-per-query, 3 = generic one-thread-per-element) on
    encoder-shaped calls: both against the oracle (2e-5), bitwise repeatable; the removed LDS variants are refused."""
    k = _kernels()
    x = _grid_inputs(9, B, shapes, jitter)
    d = {n: t.to(DEV) for n, t in x.items()}
    o1 = TA.msda_forward_variant(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 1).cpu()
    o3 = TA.msda_forward_variant(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 3).cpu()
    ref = OM.msda_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
    assert (o1 - ref).abs().max() < 2e-5 and (o3 - ref).abs().max() < 2e-5
    o1b = TA.msda_forward_variant(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 1).cpu()
    assert torch.equal(o1, o1b)
    with pytest.raises(Exception):
        TA.msda_forward_variant(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"], 13)


@pytest.mark.parametrize("shapes,B,jitter", [
    ([(19, 32), (10, 16), (5, 8), (3, 4)], 2, 0.3),
    ([(19, 32), (10, 16), (5, 8), (3, 4)], 1, 8.0),
    ([(38, 63), (19, 32), (10, 16), (5, 8)], 1, 1.0),
    ([(9, 13), (5, 7)], 3, 0.5),
])
def test_msda_backward_tile_variant(shapes, B, jitter):
    """Backward variant 2 (grad_value accumulated in LDS windows) vs the oracle and vs variant 1."""
    k = _kernels()
    x = _grid_inputs(9, B, shapes, jitter)
    d = {n: t.to(DEV) for n, t in x.items()}
    rgv, rgl, rga = OM.msda_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], x["grad_out"])
    for variant in (2, 1):
        gv, gl, ga = TA.msda_backward_variant(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"],
                                              d["grad_out"], variant)
        assert (gv.cpu() - rgv).abs().max() < 3e-4 * max(1.0, float(rgv.abs().max())), variant
        assert (ga.cpu() - rga).abs().max() < 2e-4, variant
        assert (gl.cpu() - rgl).abs().max() < 5e-3 * max(1.0, float(rgl.abs().max()) / 50), variant


@pytest.mark.parametrize("encoder_shaped", [True, False])
def test_msda_backward_out_entry_clears_grad_value_itself(encoder_shaped):
    """egtr_msda_backward_out_f32: grad_value arrives as garbage (NaN) and comes back equal to what the zero-initialised
    entry accumulates -- encoder-shaped calls clear it inside the wave-per-query kernel, others through a memset."""
    from egtr_amd import _lib, ops
    lib = _lib.lib()
    shapes = [(38, 63), (19, 32), (10, 16), (5, 8)]
    if encoder_shaped:
        x = _grid_inputs(13, 2, shapes, 0.7)
    else:
        x = W.make_msda_inputs(14, 2, 300, 8, 32, shapes, 4, oob_frac=0.2)
    d = {n: t.to(DEV) for n, t in x.items()}
    B, S, M, D = d["value"].shape
    Lq, L, P = d["loc"].shape[1], d["shapes"].shape[0], d["loc"].shape[4]
    outs = []
    for entry, fill in ((lib.egtr_msda_backward_f32, 0.0), (lib.egtr_msda_backward_out_f32, float("nan"))):
        gv = torch.full_like(d["value"], fill)
        gl = torch.full_like(d["loc"], float("nan"))
        ga = torch.full_like(d["attn"], float("nan"))
        _lib.check(entry(ops._stream(), d["grad_out"].data_ptr(), d["value"].data_ptr(), d["shapes"].data_ptr(),
                         d["lsi"].data_ptr(), d["loc"].data_ptr(), d["attn"].data_ptr(), B, S, M, D, L, Lq, P,
                         gv.data_ptr(), gl.data_ptr(), ga.data_ptr()), "msda backward")
        outs.append((gv.cpu(), gl.cpu(), ga.cpu()))
    (gv0, gl0, ga0), (gv1, gl1, ga1) = outs
    assert torch.isfinite(gv1).all()
    assert (gv0 - gv1).abs().max() < 1e-5 * max(1.0, float(gv0.abs().max()))   # atomics: order of the additions only
    assert torch.equal(gl0, gl1) and torch.equal(ga0, ga1)


def test_msda_backward_tile_variant_arbitrary_queries():
    k = _kernels()
    shapes = [(19, 32), (10, 16), (5, 8), (3, 4)]
    for Lq in (1100, 65):
        x = W.make_msda_inputs(41 + Lq, 2, Lq, 8, 32, shapes, 4, oob_frac=0.2)
        d = {n: t.to(DEV) for n, t in x.items()}
        gv, gl, ga = TA.msda_backward_variant(d["value"], d["shapes"], d["lsi"], d["loc"], d["attn"],
                                              d["grad_out"], 2)
        rgv, rgl, rga = OM.msda_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], x["grad_out"])
        assert (gv.cpu() - rgv).abs().max() < 3e-4 * max(1.0, float(rgv.abs().max()))
        assert (ga.cpu() - rga).abs().max() < 2e-4
        assert (gl.cpu() - rgl).abs().max() < 5e-3 * max(1.0, float(rgl.abs().max()) / 50)
    x = _grid_inputs(6, 1, shapes, 0.3)
    d = {n: t.to(DEV) for n, t in x.items()}
    gv, gl, ga = TA.msda_backward_variant(d["value"], d["shapes"], d["lsi"], torch.full_like(d["loc"], 3.0),
                                          d["attn"], d["grad_out"], 2)
    assert gv.abs().max().item() == 0 and gl.abs().max().item() == 0 and ga.abs().max().item() == 0


def test_msda_generic_shapes():
    for (M, D, shapes, P) in ((3, 20, [(6, 5), (2, 3)], 2), (4, 16, [(7, 9)], 3), (8, 32, [(4, 4), (2, 2), (1, 1)], 4)):
        x = W.make_msda_inputs(7, 2, 13, M, D, shapes, P)
        out, gv, gl, ga = _run_msda(x)
        ref = OM.msda_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"])
        rgv, rgl, rga = OM.msda_backward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], x["grad_out"])
        assert (out - ref).abs().max() < 2e-5 and (gv - rgv).abs().max() < 2e-4
        assert (ga - rga).abs().max() < 2e-4 and (gl - rgl).abs().max() < 5e-3


def test_msda_all_samples_out_of_range_and_exact_centres():
    shapes = [(4, 5), (2, 3), (2, 2), (1, 1)]
    x = W.make_msda_inputs(3, 1, 6, 8, 32, shapes, 4)
    x["loc"] = torch.full_like(x["loc"], 3.0)
    out, gv, gl, ga = _run_msda(x)
    assert out.abs().max() == 0 and gv.abs().max() == 0 and gl.abs().max() == 0 and ga.abs().max() == 0
    x["loc"] = torch.full_like(x["loc"], float("nan"))
    out, gv, gl, ga = _run_msda(x)
    assert out.abs().max() == 0 and gv.abs().max() == 0
    # exact pixel centres of level 0 reproduce that pixel (weights 1,0,0,0), uniform attention over 16 samples
    x = W.make_msda_inputs(3, 1, 6, 8, 32, shapes, 4)
    loc = torch.zeros_like(x["loc"])
    for l, (H, Wd) in enumerate(shapes):
        loc[..., l, :, 0] = 0.5 / Wd
        loc[..., l, :, 1] = 0.5 / H
    x["loc"], x["attn"] = loc, torch.full_like(x["attn"], 1 / 16)
    out = _run_msda(x, bwd=False)[0].reshape(1, 6, 8, 32)
    starts = [0, 20, 26, 30]
    expect = sum(x["value"][0, s] for s in starts) * 0.25
    assert (out[0, 0] - expect).abs().max() < 1e-6


def test_msda_full_size_properties():
    """BASELINE shape (600x1000 -> S = 12537, encoder Lq = S): size-independent properties + sampled rows."""
    shapes = [(75, 125), (38, 63), (19, 32), (10, 16)]
    S = sum(h * w for h, w in shapes)
    x = W.make_msda_inputs(11, 1, S, 8, 32, shapes, 4, oob_frac=0.05)
    out = _run_msda(x, bwd=False)[0]
    # (1) linearity in value
    x2 = dict(x)
    x2["value"] = 2.5 * x["value"]
    assert (_run_msda(x2, bwd=False)[0] - 2.5 * out).abs().max() < 1e-4
    # (2) constant value + all samples strictly inside => output == constant (attention weights sum to 1)
    x3 = dict(x)
    x3["value"] = torch.full_like(x["value"], 0.75)
    x3["loc"] = x["loc"].clamp(0.3, 0.7)
    assert (_run_msda(x3, bwd=False)[0] - 0.75).abs().max() < 1e-5
    # (3) a sample of queries against the oracle
    idx = torch.arange(0, S, 97)
    xs = dict(x)
    xs["loc"], xs["attn"] = x["loc"][:, idx].contiguous(), x["attn"][:, idx].contiguous()
    ref = OM.msda_forward(xs["value"], xs["shapes"], xs["lsi"], xs["loc"], xs["attn"])
    assert (out[:, idx] - ref).abs().max() < 2e-5
    # (4) backward: grad_value checksum = sum over samples of in-range bilinear weight mass (grad_out = 1)
    x["grad_out"] = torch.ones_like(x["grad_out"])
    _, gv, gl, ga = _run_msda(x)
    rgv, _, rga = OM.msda_backward(xs["value"], xs["shapes"], xs["lsi"], xs["loc"], xs["attn"],
                                   torch.ones(1, len(idx), 256))
    assert (ga[:, idx] - rga).abs().max() < 2e-4
    assert torch.isfinite(gv).all() and torch.isfinite(gl).all()


def test_msda_bf16_forward():
    k = _kernels()
    shapes = [(19, 32), (10, 16), (5, 8), (3, 4)]
    for Lq in (200, 33):
        x = W.make_msda_inputs(21, 2, Lq, 8, 32, shapes, 4)
        vb = x["value"].to(torch.bfloat16)
        out = k.ms_deform_attn_forward(vb.to(DEV), x["shapes"].to(DEV), x["lsi"].to(DEV), x["loc"].to(DEV),
                                       x["attn"].to(DEV), 64).float().cpu()
        ref = OM.msda_forward(vb.float(), x["shapes"], x["lsi"], x["loc"], x["attn"])
        assert (out - ref).abs().max() < 2e-2  # one bf16 rounding of an O(1) output


def test_msda_argument_checks():
    k = _kernels()
    x = W.make_msda_inputs(1, 1, 4, 8, 32, [(2, 2), (1, 1), (1, 1), (1, 1)], 4)
    with pytest.raises(RuntimeError):
        k.ms_deform_attn_forward(x["value"], x["shapes"], x["lsi"], x["loc"], x["attn"], 64)  # CPU tensors
    d = {n: t.to(DEV) for n, t in x.items()}
    with pytest.raises(RuntimeError):
        k.ms_deform_attn_forward(d["value"].transpose(1, 2), d["shapes"], d["lsi"], d["loc"], d["attn"], 64)


# ------------------------------------------------------------------------------------------- self-attention
def _attn_ref(q, k, v, M):
    B, N, MD = q.shape
    D = MD // M
    h = lambda t: t.view(B, N, M, D).transpose(1, 2)  # noqa: E731
    w = torch.softmax(h(q) @ h(k).transpose(-1, -2), -1)
    return (w @ h(v)).transpose(1, 2).reshape(B, N, MD)


@pytest.mark.parametrize("B,N", [(1, 200), (2, 100), (1, 37), (2, 300), (1, 16), (1, 5)])
def test_self_attention_forward_backward(B, N):
    from egtr_amd.ops import decoder_self_attention
    rng = W.rng_inputs(50 + N)
    q, k, v, go = [torch.from_numpy(rng.standard_normal((B, N, 256))).float() for _ in range(4)]
    q = q * 32 ** -0.5
    qd, kd, vd = [t.to(DEV).requires_grad_(True) for t in (q, k, v)]
    out, qh, kh = decoder_self_attention(qd, kd, vd, 8, want_maps=True)
    gqh, gkh = torch.from_numpy(rng.standard_normal((2, B, 8, N, 32))).float()
    (out * go.to(DEV)).sum().backward(retain_graph=True)
    g1 = [t.grad.clone().cpu() for t in (qd, kd, vd)]
    q64, k64, v64 = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = _attn_ref(q64, k64, v64, 8)
    (ref * go.double()).sum().backward()
    assert (out.detach().cpu() - ref.detach().float()).abs().max() < 2e-5
    assert (qh.detach().cpu() - q.view(B, N, 8, 32).transpose(1, 2)).abs().max() == 0
    assert (kh.detach().cpu() - k.view(B, N, 8, 32).transpose(1, 2)).abs().max() == 0
    for a, b in zip(g1, (q64, k64, v64)):
        assert (a - b.grad.float()).abs().max() < 5e-5 * max(1.0, float(b.grad.abs().max()))
    # gradients flowing through the retained maps fold back into q / k
    for t in (qd, kd, vd):
        t.grad = None
    ((qh * gqh.to(DEV)).sum() + (kh * gkh.to(DEV)).sum()).backward()
    assert (qd.grad.cpu() - gqh.transpose(1, 2).reshape(B, N, 256)).abs().max() < 1e-6
    assert (kd.grad.cpu() - gkh.transpose(1, 2).reshape(B, N, 256)).abs().max() < 1e-6


def test_self_attention_vs_reference_golden(golden_dir):
    """Reference MHA module fixture: q/k projections on the host, core in HIP, out_proj on the host."""
    from egtr_amd.deformable_detr import DeformableDetrMultiheadAttention
    g = Hh.load_golden(golden_dir, "mha.npz")
    shapes = json.loads(str(g["shapes"]))
    m = DeformableDetrMultiheadAttention(256, 8)
    m.load_state_dict(W.fill_state_dict(shapes, seed=int(g["seed"])))
    m = m.to(DEV).eval()
    with torch.no_grad():
        o, _, q, k = m(_t(g["x"]).to(DEV), position_embeddings=_t(g["pos"]).to(DEV), output_attention_states=True)
    assert (o.cpu() - _t(g["out"])).abs().max() < 2e-5
    assert (q.cpu() - _t(g["q"])).abs().max() < 1e-5 and (k.cpu() - _t(g["k"])).abs().max() < 1e-5


def test_self_attention_mask_map_and_dropout_options(golden_dir):
    """dd:1198-1237: padding mask + output_attentions=True against the reference module's fixture; without a mask the
    explicit route equals the fused kernel; attention dropout in training runs and keeps the expectation."""
    from egtr_amd.deformable_detr import DeformableDetrMultiheadAttention
    g = Hh.load_golden(golden_dir, "mha.npz")
    shapes = json.loads(str(g["shapes"]))
    m = DeformableDetrMultiheadAttention(256, 8)
    m.load_state_dict(W.fill_state_dict(shapes, seed=int(g["seed"])))
    m = m.to(DEV).eval()
    x, pos = _t(g["x"]).to(DEV), _t(g["pos"]).to(DEV)
    with torch.no_grad():
        o, w, _, _ = m(x, attention_mask=_t(g["mask"]).to(DEV), position_embeddings=pos, output_attentions=True)
        o_fused, w_none, _, _ = m(x, position_embeddings=pos)
        o_map, w_map, _, _ = m(x, position_embeddings=pos, output_attentions=True)
    assert w_none is None and w.shape == (2, 8, 37, 37)
    assert (o.cpu() - _t(g["out_masked"])).abs().max() < 2e-5
    assert (w.cpu() - _t(g["attn_masked"])).abs().max() < 1e-6
    assert (o_map - o_fused).abs().max() < 2e-5 and (w_map.sum(-1) - 1).abs().max() < 1e-5
    md = DeformableDetrMultiheadAttention(256, 8, dropout=0.25)
    md.load_state_dict(m.state_dict())
    md = md.to(DEV).train()
    torch.manual_seed(0)
    outs = torch.stack([md(x, position_embeddings=pos)[0].detach() for _ in range(200)])
    assert (outs[0] - outs[1]).abs().max() > 1e-3  # dropout is live
    assert (outs.mean(0) - o_fused).abs().max() < 0.25 * float(o_fused.abs().max())  # E[dropout(p) v] = p v
    md.eval()
    with torch.no_grad():
        assert (md(x, position_embeddings=pos)[0] - o_fused).abs().max() == 0  # eval: the fused kernel again


# --------------------------------------------------------------------------------------------- relation head
def _head_inputs(seed, B, N, T, R, C):
    rng = W.rng_inputs(seed)
    r = lambda *s, sc=1.0: torch.from_numpy(rng.standard_normal(s) * sc).float()  # noqa: E731
    d = dict(gate_q=r(B, N, T), gate_k=r(B, N, T), uq=r(B, N, T, 512, sc=0.5), uk=r(B, N, T, 512, sc=0.5),
             b1=r(512, sc=0.1), w2r=r(256, 256, sc=1 / 16), b2r=r(256, sc=0.1), w3r=r(R, 256, sc=1 / 16),
             b3r=r(R, sc=0.1), w2c=r(256, 256, sc=1 / 16), b2c=r(256, sc=0.1), w3c=r(1, 256, sc=1 / 16),
             b3c=r(1, sc=0.1))
    trip = r(C + 1, C + 1, R)
    node = torch.from_numpy(rng.integers(0, C + 1, (B, N))).long()
    return d, trip, node


@pytest.mark.parametrize("B,N,T,R", [(1, 200, 7, 50), (2, 24, 4, 7), (1, 100, 4, 30), (1, 33, 9, 64), (1, 7, 1, 1)])
def test_relation_head_forward(B, N, T, R):
    import cpu_kernels as ck
    from egtr_amd.ops import relation_head
    d, trip, node = _head_inputs(60 + N, B, N, T, R, 11)
    dd = {k: v.to(DEV) for k, v in d.items()}
    rel, conn, gm = relation_head(*dd.values(), trip.to(DEV), node.to(DEV), True)
    d64 = {k: v.double() for k, v in d.items()}
    rrel, rconn, rgm = ck.relation_head(*d64.values(), trip.double(), node, True)
    assert (rel.cpu() - rrel.float()).abs().max() < 2e-4
    assert (conn.cpu() - rconn.float()).abs().max() < 2e-4
    assert (gm.cpu() - rgm.float()).abs().max() < 1e-5
    rel2, conn2, gm2 = relation_head(*dd.values(), None, None, False)  # no frequency bias
    rrel2, _, _ = ck.relation_head(*d64.values(), None, None, False)
    assert (rel2.cpu() - rrel2.float()).abs().max() < 2e-4 and gm2 is None


@pytest.mark.parametrize("B,N,T,R", [(1, 200, 7, 50), (2, 24, 4, 7), (1, 100, 4, 30), (1, 33, 9, 64), (1, 7, 1, 1),
                                     (2, 37, 7, 32)])
def test_relation_head_split_bf16_is_fp32_accurate(B, N, T, R):
    """The inference kernel evaluates layers 2 / 3 on the bf16 matrix cores from three-way bf16 splits of both operands
    (six cross terms, fp32 accumulation).  Claim under test: the result is an fp32 result -- against the float64
    restatement its error is the same size as the error of the exact-f32 MFMA kernel (same 2e-4 bar as
    test_relation_head_forward, and within 2.5x of that kernel's own max / Frobenius error)."""
    import cpu_kernels as ck
    from egtr_amd import ops
    d, trip, node = _head_inputs(60 + N, B, N, T, R, 11)
    dd = {k: v.to(DEV) for k, v in d.items()}
    d64 = {k: v.double() for k, v in d.items()}
    rrel, rconn, rgm = ck.relation_head(*d64.values(), trip.double(), node, True)
    rel32, conn32, _ = ops.RelationHeadFunction.apply(*dd.values(), trip.to(DEV), node.to(DEV), True)
    w2xr, w3xr, w2xc = ops.rel_head_split_weights(dd["w2r"], dd["w3r"], dd["w2c"])
    rel, conn, gm = ops.relation_head_split_bf16(
        dd["gate_q"], dd["gate_k"], dd["uq"], dd["uk"], dd["b1"], w2xr, dd["b2r"], w3xr, dd["b3r"], w2xc, dd["b2c"],
        dd["w3c"], dd["b3c"], R, trip.to(DEV), node.to(DEV), True)
    for got, got32, ref in ((rel, rel32, rrel), (conn, conn32, rconn)):
        e = (got.cpu().double() - ref).abs()
        e32 = (got32.cpu().double() - ref).abs()
        assert e.max() < 2e-4
        assert e.max() <= 2.5 * e32.max() + 1e-6, (float(e.max()), float(e32.max()))
        assert e.norm() <= 2.5 * e32.norm() + 1e-6, (float(e.norm()), float(e32.norm()))
    assert (gm.cpu() - rgm.float()).abs().max() < 1e-5
    rel2, _, gm2 = ops.relation_head_split_bf16(
        dd["gate_q"], dd["gate_k"], dd["uq"], dd["uk"], dd["b1"], w2xr, dd["b2r"], w3xr, dd["b3r"], w2xc, dd["b2c"],
        dd["w3c"], dd["b3c"], R, None, None, False)
    rrel2, _, _ = ck.relation_head(*d64.values(), None, None, False)
    assert (rel2.cpu().double() - rrel2).abs().max() < 2e-4 and gm2 is None
    # sigmoid in the epilogue == sigmoid of the logits
    rel3, conn3, _ = ops.relation_head_split_bf16(
        dd["gate_q"], dd["gate_k"], dd["uq"], dd["uk"], dd["b1"], w2xr, dd["b2r"], w3xr, dd["b3r"], w2xc, dd["b2c"],
        dd["w3c"], dd["b3c"], R, trip.to(DEV), node.to(DEV), False, sigmoid=True)
    assert (rel3 - rel.sigmoid()).abs().max() < 1e-6 and (conn3 - conn.sigmoid()).abs().max() < 1e-6


@pytest.mark.parametrize("M,K,N,relu", [(12537, 256, 256, False), (12537, 256, 1024, True), (12537, 1024, 256, False),
                                        (5000, 256, 384, False), (4099, 32, 128, True), (129, 64, 128, False)])
def test_linear_split_bf16_is_fp32_accurate(M, K, N, relu):
    """Token-sized linear on the bf16 matrix cores from three-way operand splits (csrc/gemm_split.hip) against float64:
    error of the same size as the vendor fp32 GEMM's (within 2.5x in max and Frobenius norm), incl. a row-strided input,
    a partial last row tile and the ReLU epilogue."""
    from egtr_amd import ops
    rng = W.rng_inputs(700 + M + K)
    x = torch.from_numpy(rng.standard_normal((M, K + 8))).float()
    w = torch.from_numpy(rng.standard_normal((N, K)) / np.sqrt(K)).float()
    b = torch.from_numpy(rng.standard_normal(N) * 0.1).float()
    xd = x.to(DEV)[:, :K]           # row stride K + 8: a column block of a wider buffer
    wt = ops.gemm_split_weights(w.to(DEV))
    y = ops.linear_split_bf16(xd, wt, b.to(DEV), N, relu=relu)
    ref = x[:, :K].double() @ w.double().t() + b.double()
    y32 = torch.nn.functional.linear(xd.contiguous(), w.to(DEV), b.to(DEV))
    if relu:
        ref, y32 = ref.clamp_min(0), y32.clamp_min(0)
    e = (y.cpu().double() - ref).abs()
    e32 = (y32.cpu().double() - ref).abs()
    assert y.shape == (M, N)
    assert e.max() < 1e-5 * max(1.0, float(ref.abs().max()))
    assert e.max() <= 2.5 * e32.max() + 1e-7, (float(e.max()), float(e32.max()))
    assert e.norm() <= 2.5 * e32.norm() + 1e-7, (float(e.norm()), float(e32.norm()))
    out = torch.full((M, N), 7.0, device=DEV)
    y2 = ops.linear_split_bf16(xd, wt, None, N, relu=False, out=out)     # no bias, caller's buffer
    assert y2.data_ptr() == out.data_ptr()
    assert (out.cpu().double() - x[:, :K].double() @ w.double().t()).abs().max() < 1e-5 * max(1.0, float(ref.abs().max()))


def test_linear_split_bf16_grouped_equals_single_launches():
    """Several products in one launch (shared grid) give bit-identical results to one launch each."""
    from egtr_amd import ops
    rng = W.rng_inputs(321)
    M, K = 12537, 256
    xa = torch.from_numpy(rng.standard_normal((M, K))).float().to(DEV)
    xb = torch.from_numpy(rng.standard_normal((M, K))).float().to(DEV)
    ws = [torch.from_numpy(rng.standard_normal((n, K)) / 16).float().to(DEV) for n in (256, 384, 256)]
    bs = [torch.from_numpy(rng.standard_normal(n) * 0.1).float().to(DEV) for n in (256, 384, 256)]
    wts = [ops.gemm_split_weights(w) for w in ws]
    single = [ops.linear_split_bf16(x, wt, b, w.shape[0], relu=r)
              for x, wt, b, w, r in ((xa, wts[0], bs[0], ws[0], False), (xb, wts[1], None, ws[1], True),
                                     (xa, wts[2], bs[2], ws[2], False))]
    out2 = torch.empty(M, 256, device=DEV)
    grouped = ops.linear_split_bf16_grouped([dict(x=xa, wt=wts[0], N=256, b=bs[0]),
                                             dict(x=xb, wt=wts[1], N=384, relu=True),
                                             dict(x=xa, wt=wts[2], N=256, b=bs[2], out=out2)])
    for a, b in zip(single, grouped):
        assert torch.equal(a, b)
    assert grouped[2].data_ptr() == out2.data_ptr()


def test_relation_head_split_weights_sum_to_the_fp32_weights():
    """hi + mid + lo reproduces every fp32 weight to <= 2^-24 relative, incl. large / tiny / denormal-range values."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(256, 256, generator=g) * torch.logspace(-12, 6, 256)[:, None]
    p = ops._split3_bf16(w.to(DEV)).float().cpu()
    rec = (p[0].double() + p[1].double() + p[2].double()).float()
    assert ((rec - w).abs() <= w.abs() * 2.0 ** -23).all()


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("B,N,T,R", [(1, 200, 7, 50), (2, 24, 4, 7), (1, 33, 9, 64), (1, 7, 1, 1), (2, 300, 9, 50),
                                     (3, 13, 10, 33)])
def test_relation_head_forward_bf16_matrix_cores(B, N, T, R, packed, monkeypatch):
    """bf16-weight forward against the fp64 restatement evaluated with the same bf16-rounded weights AND bf16-rounded
    per-query tables (a bf16 model produces uq / uk in bf16; both kernels get the rounded values, the packed one as a bf16
    tensor).  packed: all three layers on v_mfma_f32_32x32x16_bf16 (rel_head_fwd_bf16p: tables in operand order, gates
    rounded to bf16 as an operand, ragged 4 x 8 pair tiles at N = 33 / 7 / 13 / 300); not packed: fp32 VALU layer 1
    (rel_head_fwd_bf16w).  What differs from the restatement is the bf16 rounding of operands: tolerance 3e-2 absolute on
    logits of magnitude ~1-4 and 5e-3 relative Frobenius (a fragment-layout mistake is O(1))."""
    import cpu_kernels as ck
    from egtr_amd import ops
    from egtr_amd.ops import relation_head_bf16w
    monkeypatch.setattr(ops, "REL_HEAD_BF16_PACKED", packed)
    d, trip, node = _head_inputs(160 + N, B, N, T, R, 11)
    wnames = ("w2r", "w3r", "w2c", "w3c")
    d["uq"], d["uk"] = d["uq"].bfloat16().float(), d["uk"].bfloat16().float()
    dd = {k: (v.to(DEV).bfloat16() if k in wnames or (packed and k in ("uq", "uk")) else v.to(DEV)) for k, v in d.items()}
    rel, conn, gm = relation_head_bf16w(*dd.values(), trip.to(DEV), node.to(DEV), True)
    d64 = {k: (v.bfloat16().double() if k in wnames else v.double()) for k, v in d.items()}
    rrel, rconn, rgm = ck.relation_head(*d64.values(), trip.double(), node, True)
    for got, ref in ((rel.cpu().double(), rrel), (conn.cpu().double(), rconn)):
        assert (got - ref).abs().max() < 3e-2
        assert (got - ref).norm() / ref.norm() < 5e-3
    assert (gm.cpu() - rgm.float()).abs().max() < 1e-5
    rel2, _, gm2 = relation_head_bf16w(*dd.values(), None, None, False)
    rrel2, _, _ = ck.relation_head(*d64.values(), None, None, False)
    assert (rel2.cpu().double() - rrel2).abs().max() < 3e-2 and gm2 is None


def test_relation_head_streams_kernel_equals_torch_composition():
    """egtr_rel_head_streams_f32 (one launch; rebuilt every training step) == ops.rel_head_split_weights, bit for bit."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(5)
    for R in (50, 7, 64):
        w2r = (torch.randn(256, 256, generator=g) * torch.logspace(-6, 3, 256)[None]).to(DEV)
        w2c = torch.randn(256, 256, generator=g).to(DEV)
        w3r = torch.randn(R, 256, generator=g).to(DEV)
        a = ops.rel_head_split_weights(w2r, w3r, w2c)
        b = ops.rel_head_streams(w2r, w3r, w2c)
        for x, y in zip(a, b):
            assert x.shape == y.shape and torch.equal(x.view(torch.int16), y.view(torch.int16))


@pytest.mark.parametrize("B,N,T,R", [(2, 24, 4, 7), (1, 37, 7, 50)])
def test_relation_head_training_forward_on_split_arithmetic_vs_exact(B, N, T, R, monkeypatch):
    """ops.RelationHeadFunction with its forward on the split-bf16 kernel (ops.REL_HEAD_TRAIN_X6, the default) against the
    exact-f32 kernel: logits to fp32 rounding, every gradient within 1e-4 of its scale (a hidden-2 unit within rounding
    of zero may take the other side of its ReLU)."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(31 + N)
    Hd, C1 = 256, 11
    def mk(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(DEV)
    base = dict(gate_q=mk(B, N, T), gate_k=mk(B, N, T), uq=mk(B, N, T, 2 * Hd, scale=0.3), uk=mk(B, N, T, 2 * Hd, scale=0.3),
                b1=mk(2 * Hd, scale=0.1), w2r=mk(Hd, Hd, scale=Hd ** -0.5), b2r=mk(Hd, scale=0.1), w3r=mk(R, Hd, scale=Hd ** -0.5),
                b3r=mk(R, scale=0.1), w2c=mk(Hd, Hd, scale=Hd ** -0.5), b2c=mk(Hd, scale=0.1), w3c=mk(1, Hd, scale=Hd ** -0.5),
                b3c=mk(1, scale=0.1))
    trip = mk(C1, C1, R, scale=0.5)
    node = torch.randint(0, C1, (B, N), generator=g).to(DEV)
    gr, gc = mk(B, N, N, R), mk(B, N, N, 1)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "REL_HEAD_TRAIN_X6", flag)
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        rel, conn, gm = ops.RelationHeadFunction.apply(*[t[k] for k in base], trip, node, True)
        (rel * gr).sum().add((conn * gc).sum()).backward()
        res[flag] = (rel.detach(), conn.detach(), gm.detach(), {k: v.grad for k, v in t.items()})
    (r1, c1_, g1, d1), (r0, c0, g0, d0) = res[True], res[False]
    assert (r1 - r0).abs().max() < 2e-5 * max(1.0, float(r0.abs().max()))
    assert (c1_ - c0).abs().max() < 2e-5 * max(1.0, float(c0.abs().max()))
    assert (g1 - g0).abs().max() < 1e-5
    for k in base:
        scale = max(1e-3, float(d0[k].abs().max()))
        assert (d1[k] - d0[k]).abs().max() < 1e-4 * scale, k


@pytest.mark.parametrize("B,N,T,R", [(2, 12, 4, 7), (1, 200, 7, 50), (3, 40, 7, 50), (1, 33, 9, 64), (2, 7, 1, 1),
                                     (1, 300, 9, 50)])
def test_relation_head_backward_matches_autograd(B, N, T, R):
    """Forward that saves h1 / h2 + backward = rocBLAS GEMMs on the saved activations + the HIP pairwise kernels
    (egtr_rel_head_backward_pairs_f32), against fp64 autograd through the reference formulation (tests/cpu_kernels)."""
    import cpu_kernels as ck
    from egtr_amd.ops import relation_head
    d, trip, node = _head_inputs(77 + N, B, N, T, R, 5)
    dd = {k: v.to(DEV).requires_grad_(True) for k, v in d.items()}
    rel, conn, _ = relation_head(*dd.values(), trip.to(DEV), node.to(DEV), False)
    rng = W.rng_inputs(78)
    g1 = torch.from_numpy(rng.standard_normal(tuple(rel.shape))).float()
    g2 = torch.from_numpy(rng.standard_normal(tuple(conn.shape))).float()
    ((rel * g1.to(DEV)).sum() + (conn * g2.to(DEV)).sum()).backward()
    d64 = {k: v.double().requires_grad_(True) for k, v in d.items()}
    rrel, rconn, _ = ck.relation_head(*d64.values(), trip.double(), node, False)
    ((rrel * g1.double()).sum() + (rconn * g2.double()).sum()).backward()
    for k in d:
        got, want = dd[k].grad.cpu().double(), d64[k].grad
        if B * N * N <= 4096:
            assert (got - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max())), k
        else:
            # Large problems: of the B*N*N*512 ReLU inputs a handful lie within fp32 rounding of zero, and the fp32
            # kernel and the fp64 reference then disagree on that mask bit (an O(1) change of ONE term of a sum over
            # 1e5 pairs).  The relative Frobenius error is insensitive to those few terms; 3e-3 still catches any
            # systematic error (a wrong term, a missing slot, a transposed weight are all > 1e-1).
            assert float((got - want).norm() / want.norm().clamp_min(1e-12)) < 3e-3, k


# ---------------------------------------------------------------------------------------------- skinny linear
@pytest.mark.parametrize("M,K,N", [(200, 256, 256), (200, 256, 1024), (200, 1024, 256), (400, 256, 150),
                                   (1400, 256, 513), (7, 64, 2), (300, 512, 50), (33, 320, 4), (1, 256, 1),
                                   (800, 256, 128), (13, 64, 64), (801, 256, 512), (3, 1024, 64)])
def test_skinny_linear(M, K, N):
    from egtr_amd.ops import linear
    rng = W.rng_inputs(K + N + M)
    x = torch.from_numpy(rng.standard_normal((M, K))).float()
    w = torch.from_numpy(rng.standard_normal((N, K)) / K ** 0.5).float()
    b = torch.from_numpy(rng.standard_normal((N,))).float()
    for (alpha, relu, bias) in ((1.0, False, True), (0.1767767, False, True), (1.0, True, True), (1.0, False, False)):
        xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        bd = b.to(DEV).requires_grad_(True) if bias else None
        y = linear(xd, wd, bd, alpha=alpha, relu=relu)
        x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
        b64 = b.double().requires_grad_(True) if bias else None
        r = torch.nn.functional.linear(x64, w64, b64) * alpha
        r = torch.relu(r) if relu else r
        assert (y.detach().cpu() - r.detach().float()).abs().max() < 2e-5 * max(1.0, float(r.abs().max()))
        go = torch.from_numpy(rng.standard_normal((M, N))).float()
        (y * go.to(DEV)).sum().backward()
        (r * go.double()).sum().backward()
        assert (xd.grad.cpu() - x64.grad.float()).abs().max() < 1e-4 * max(1.0, float(x64.grad.abs().max()))
        assert (wd.grad.cpu() - w64.grad.float()).abs().max() < 1e-4 * max(1.0, float(w64.grad.abs().max()))
        if bias:
            assert (bd.grad.cpu() - b64.grad.float()).abs().max() < 1e-4 * max(1.0, float(b64.grad.abs().max()))


def test_skinny_linear_batched_input_and_determinism():
    from egtr_amd.ops import linear
    x = torch.randn(2, 200, 7, 256, device=DEV)
    w = torch.randn(513, 256, device=DEV) / 16
    y1, y2 = linear(x, w), linear(x, w)
    assert y1.shape == (2, 200, 7, 513) and torch.equal(y1, y2)
    assert (y1 - torch.nn.functional.linear(x.double(), w.double()).float()).abs().max() < 2e-5 * float(y1.abs().max())


def test_add_layernorm_and_bias_act():
    from egtr_amd import ops
    rng = W.rng_inputs(3)
    x = torch.from_numpy(rng.standard_normal((3, 77, 256))).float()
    r = torch.from_numpy(rng.standard_normal((3, 77, 256))).float()
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(1 + 0.1 * rng.standard_normal(256)).float())
        ln.bias.copy_(torch.from_numpy(0.1 * rng.standard_normal(256)).float())
    ref = torch.nn.functional.layer_norm((x + r).double(), (256,), ln.weight.double(), ln.bias.double(), ln.eps)
    lnd = ln.to(DEV)
    xd, rd = x.to(DEV).requires_grad_(True), r.to(DEV).requires_grad_(True)
    y = ops.add_layer_norm(xd, rd, lnd)
    assert (y.detach().cpu() - ref.float()).abs().max() < 2e-5
    go = torch.from_numpy(rng.standard_normal((3, 77, 256))).float()
    (y * go.to(DEV)).sum().backward()
    x64 = (x + r).double().requires_grad_(True)
    w64, b64 = ln.weight.detach().cpu().double().requires_grad_(True), ln.bias.detach().cpu().double().requires_grad_(True)
    (torch.nn.functional.layer_norm(x64, (256,), w64, b64, ln.eps) * go.double()).sum().backward()
    assert (xd.grad.cpu() - x64.grad.float()).abs().max() < 1e-4 and torch.equal(xd.grad, rd.grad)
    assert (lnd.weight.grad.cpu() - w64.grad.float()).abs().max() < 1e-3
    # bias + residual + relu on NCHW, aligned (HW % 4 == 0) and unaligned shapes
    for (n, c, h, w_) in ((2, 64, 10, 12), (1, 7, 5, 9), (1, 256, 75, 125), (2, 24, 38, 63), (3, 5, 2, 3), (1, 3, 150, 250)):
        a = torch.from_numpy(rng.standard_normal((n, c, h, w_))).float()
        res = torch.from_numpy(rng.standard_normal((n, c, h, w_))).float()
        b = torch.from_numpy(rng.standard_normal((c,))).float()
        for use_res, relu in ((True, True), (False, True), (False, False)):
            got = ops.bias_act_(a.to(DEV).clone(), b.to(DEV), res.to(DEV) if use_res else None, relu=relu).cpu()
            exp = a + b.view(1, -1, 1, 1) + (res if use_res else 0)
            exp = torch.relu(exp) if relu else exp
            assert torch.equal(got, exp) or (got - exp).abs().max() < 1e-6


@pytest.mark.parametrize("rows", [8193, 12537])
def test_add_layernorm_token_sized(rows):
    """Token-sized inputs (encoder: 12 537 rows; an odd count that is not a multiple of the 4 rows per workgroup): plain
    and "+ pos" variants against fp64 LayerNorm."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g)
    r = torch.randn(rows, 256, generator=g)
    pos = torch.randn(rows, 256, generator=g)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
        ln.bias.copy_(0.2 * torch.randn(256, generator=g))
    want = torch.nn.functional.layer_norm((x + r).double(), (256,), ln.weight.double(), ln.bias.double(), ln.eps)
    lnd = ln.to(DEV)
    with torch.no_grad():
        y = ops.add_layer_norm(x.to(DEV), r.to(DEV), lnd)
        y2, yp = ops.add_layer_norm_pos(x.to(DEV), r.to(DEV), lnd, pos.to(DEV))
    assert (y.cpu() - want.float()).abs().max() < 2e-5
    assert torch.equal(y2, y)
    assert (yp.cpu() - (want + pos.double()).float()).abs().max() < 2e-5


@pytest.mark.parametrize("N,C,H,W_", [(1, 64, 300, 500), (2, 7, 33, 41), (1, 3, 8, 130), (3, 5, 1, 1)])
def test_bias_relu_maxpool_is_bitwise_maxpool_of_relu(N, C, H, W_):
    """egtr_bias_relu_maxpool3x3s2_f32 == max_pool2d(relu(x + b), 3, 2, 1) bit for bit (odd / even sizes, borders)."""
    import torch.nn.functional as F
    from egtr_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = torch.randn(N, C, H, W_, generator=g)
    b = torch.randn(C, generator=g)
    want = F.max_pool2d(torch.relu(x + b.view(1, C, 1, 1)), 3, 2, 1)
    got = ops.bias_relu_maxpool(x.to(DEV), b.to(DEV)).cpu()
    assert got.shape == want.shape
    assert torch.equal(got, want)


@pytest.mark.parametrize("RD", [2, 4])
def test_box_decode_matches_reference_composition(RD):
    """egtr_box_decode_argmax_f32 against sigmoid(delta + inverse_sigmoid(reference)) level by level (egtr:286-305), with
    reference values at and beyond the clamp points 0, 1, eps."""
    from egtr_amd import ops
    from egtr_amd.deformable_detr import inverse_sigmoid
    g = torch.Generator().manual_seed(40 + RD)
    B, Ld, N = 2, 3, 37
    delta = 2.0 * torch.randn(B, Ld, N, 4, generator=g)
    init = torch.rand(B, N, RD, generator=g)
    inter = torch.rand(B, Ld, N, RD, generator=g)
    init[0, :6, 0] = torch.tensor([0.0, 1.0, -0.3, 1.7, 1e-6, 1 - 1e-6])
    inter[1, 1, :4, -1] = torch.tensor([0.0, 1.0, 5e-6, 2.0])
    want = []
    for lvl in range(Ld):
        r = inverse_sigmoid(init if lvl == 0 else inter[:, lvl - 1])
        d = delta[:, lvl]
        want.append((d + r if RD == 4 else torch.cat([d[..., :2] + r, d[..., 2:]], -1)).sigmoid())
    want = torch.stack(want, 1)
    got = ops.box_decode(delta.to(DEV), init.to(DEV), inter.to(DEV)).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-6


def test_sine_position_embedding_matches_reference_formula():
    from egtr_amd.deformable_detr import DeformableDetrSinePositionEmbedding
    pe = DeformableDetrSinePositionEmbedding(128, normalize=True)
    for (B, H, Wd) in ((1, 75, 125), (2, 19, 32), (1, 10, 16)):
        mask = torch.ones(B, H, Wd, dtype=torch.bool)
        if B > 1:
            mask[1, H - 5:, :] = False
            mask[1, :, Wd - 7:] = False
        ref = pe(torch.zeros(B, 3, H, Wd), mask)          # CPU: the reference's elementwise formula
        got = pe(torch.zeros(B, 3, H, Wd, device=DEV), mask.to(DEV)).cpu()
        assert got.shape == ref.shape == (B, 256, H, Wd)
        valid = mask[:, None].expand_as(ref)
        # padded positions evaluate sin/cos of ~1e6-sized arguments (ill-conditioned, never used): compare valid ones
        assert (got - ref)[valid].abs().max() < 2e-5


@pytest.mark.parametrize("geom", ["small", "multi_block"])
@pytest.mark.parametrize("mask_dtype", [torch.long, torch.bool])
def test_level_geometry_matches_reference_composition(mask_dtype, geom):
    """egtr_level_geometry_f32 against the PyTorch composition the reference uses (dd:2195-2278, 1616-1648, 850-876):
    nearest-resized masks, sine position embeddings + level_embed, valid ratios, encoder reference points; one fully
    valid image and two padded ones (valid regions 37x61 and 50x13 of 64x80)."""
    import math
    import torch.nn.functional as F
    from egtr_amd import ops
    from egtr_amd.deformable_detr import DeformableDetrEncoder, DeformableDetrSinePositionEmbedding
    if geom == "small":
        B, H, W_ = 3, 64, 80
        shapes = [(8, 10), (4, 5), (2, 3), (1, 2)]
        cuts = ((37, 61), (50, 13))
    else:  # S = 1424 tokens: several workgroups of the mask-resize kernel, a ragged last bit-mask word
        B, H, W_ = 3, 200, 333
        shapes = [(25, 42), (13, 21), (7, 11), (4, 6)]
        cuts = ((131, 290), (180, 77))
    pm = torch.ones(B, H, W_, dtype=torch.long)
    pm[1, cuts[0][0]:, :] = 0
    pm[1, :, cuts[0][1]:] = 0
    pm[2, cuts[1][0]:, :] = 0
    pm[2, :, cuts[1][1]:] = 0
    g = torch.Generator().manual_seed(3)
    level_embed = torch.randn(4, 256, generator=g)
    pe = DeformableDetrSinePositionEmbedding(128, normalize=True)
    masks = [F.interpolate(pm[None].float(), size=hw).to(torch.bool)[0] for hw in shapes]
    pos = [pe(torch.zeros(B, 1, *hw), m) for hw, m in zip(shapes, masks)]
    want_mask = torch.cat([m.flatten(1) for m in masks], 1)
    want_pos = torch.cat([p.flatten(2).transpose(1, 2) + level_embed[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
    vr = torch.stack([torch.stack([m[:, 0, :].sum(1).float() / m.shape[2], m[:, :, 0].sum(1).float() / m.shape[1]], -1)
                      for m in masks], 1)
    want_ref = DeformableDetrEncoder.get_reference_points(shapes, vr, "cpu")
    mask, posf, vrf, ref, _bits = ops.level_geometry(pm.to(mask_dtype).to(DEV), shapes, level_embed.to(DEV), 128, 10000,
                                              2 * math.pi)
    assert torch.equal(mask.cpu(), want_mask)
    bits = _bits.cpu().long() & 0xFFFFFFFF
    S_ = want_mask.shape[1]
    unpacked = ((bits[:, torch.arange(S_) // 32] >> (torch.arange(S_) % 32)) & 1).bool()
    assert torch.equal(unpacked, want_mask)
    assert torch.equal(vrf.cpu(), vr)
    assert (ref.cpu() - want_ref).abs().max() < 1e-6
    valid = want_mask[..., None].expand_as(want_pos)
    assert (posf.cpu() - want_pos)[valid].abs().max() < 2e-5
    # padded tokens: the embedding takes sin / cos of arguments up to ~1e6 (cumsum / eps); compare loosely in the
    # argument domain by checking the values stay in [-1, 1] + level_embed range
    per_level = torch.tensor([h * w for h, w in shapes])
    assert (posf.cpu() - level_embed.repeat_interleave(per_level, 0)[None]).abs().max() <= 1.0 + 1e-6


def test_linear_grouped_matches_torch():
    """egtr_linear_grouped_f32: mixed shapes, input scale, output scale, ReLU, strided output rows, no bias."""
    import torch.nn.functional as F
    from egtr_amd import ops
    g = torch.Generator().manual_seed(5)
    K = 256
    specs = [(200, 256, 1.0, 0.17677669, False, True), (200, 256, 1.0, 1.0, False, True),
             (37, 513, 5.656854, 1.0, False, False), (1400, 1, 1.0, 1.0, False, True), (64, 128, 1.0, 1.0, True, True)]
    items, want = [], []
    buf = torch.zeros(200, 3, 256, device=DEV)
    for i, (M, N, ax, al, relu, has_b) in enumerate(specs):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / 16
        b = torch.randn(N, generator=g) if has_b else None
        r = F.linear(x * ax, w, b) * al
        want.append(torch.relu(r) if relu else r)
        it = dict(x=x.to(DEV), w=w.to(DEV), b=b.to(DEV) if has_b else None, alpha_x=ax, alpha=al, relu=relu)
        if i == 1:
            it["out"] = buf[:, 1, :]
        items.append(it)
    with torch.no_grad():
        outs = ops.linear_grouped(items)
    for o, r in zip(outs, want):
        assert (o.cpu() - r).abs().max() < 2e-5 * max(1.0, float(r.abs().max()))
    assert torch.equal(outs[1], buf[:, 1, :]) and buf[:, 0].abs().max() == 0 and buf[:, 2].abs().max() == 0


def test_add_layer_norm_pos_and_bias_mask_rows():
    from egtr_amd import ops
    g = torch.Generator().manual_seed(6)
    x, res, pos = torch.randn(3, 50, 256, generator=g), torch.randn(3, 50, 256, generator=g), torch.randn(50, 256, generator=g)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(256, generator=g))
        ln.bias.copy_(torch.randn(256, generator=g))
        want = ln(x + res)
        y, yp = ops.add_layer_norm_pos(x.to(DEV), res.to(DEV), ln.to(DEV), pos.to(DEV))
    assert (y.cpu() - want).abs().max() < 2e-5
    assert (yp.cpu() - (want + pos)).abs().max() < 2e-5
    v = torch.randn(4, 77, 256, generator=g)
    b = torch.randn(4, 256, generator=g)
    keep = torch.rand(77, generator=g) > 0.3
    want = torch.where(keep[None, :, None], v + b[:, None, :], torch.zeros(()))
    got = ops.bias_mask_rows_(v.to(DEV).clone(), b.to(DEV), keep.to(DEV))
    assert torch.equal(got.cpu(), want)
    got = ops.bias_mask_rows_(v.to(DEV).clone(), b.to(DEV), None)
    assert torch.equal(got.cpu(), v + b[:, None, :])


def test_input_proj_groupnorm_flatten_matches_torch():
    """Conv bias + GroupNorm(32) + flatten(2).transpose(1, 2) + cat over four levels (dd:2209-2262) in two launches."""
    import torch.nn as nn
    from egtr_amd import ops
    torch.manual_seed(7)
    shapes = [(19, 32), (10, 16), (5, 8), (3, 3)]
    projs = nn.ModuleList([nn.Sequential(nn.Conv2d(8, 256, 1), nn.GroupNorm(32, 256)) for _ in shapes])
    with torch.no_grad():
        for p in projs:
            p[0].bias.normal_()
            p[1].weight.normal_()
            p[1].bias.normal_()
    xs = [3.0 * torch.randn(2, 256, h, w) + 1.5 for h, w in shapes]
    with torch.no_grad():
        want = torch.cat([p[1](x + p[0].bias.view(1, -1, 1, 1)).flatten(2).transpose(1, 2) for p, x in zip(projs, xs)], 1)
        got = ops.input_proj_groupnorm_flatten([x.to(DEV) for x in xs], projs.to(DEV))
    assert got.shape == want.shape
    assert (got.cpu() - want).abs().max() < 2e-5


def test_msda_fused_strided_inputs_and_keep_mask():
    """Fused entry with offsets / logits as column blocks of one [B, Lq, 384] tensor and an in-kernel padding mask,
    against masking the value rows on the host."""
    k = _kernels()
    g = torch.Generator().manual_seed(12)
    shapes = [(19, 32), (10, 16), (5, 8), (3, 4)]
    B, S = 2, sum(h * w for h, w in shapes)
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, 8, 32, generator=g)
    both = torch.randn(B, S, 384, generator=g) * 2
    ref = torch.rand(B, S, 4, 2, generator=g)
    keep = torch.rand(B, S, generator=g) > 0.25
    d = [t.to(DEV) for t in (value, shp, lsi, both, ref, keep)]
    off = d[3][..., :256].view(B, S, 8, 4, 4, 2)
    logits = d[3][..., 256:].view(B, S, 8, 16)
    out, _ = k.ms_deform_attn_forward_fused(d[0], d[1], d[2], off, logits, d[4], False, d[5])
    vm = torch.where(d[5][..., None, None], d[0], torch.zeros((), device=DEV))
    want, _ = k.ms_deform_attn_forward_fused(vm, d[1], d[2], off.contiguous(), logits.contiguous(), d[4], False, None)
    assert (out - want).abs().max().item() < 1e-6
    # the same mask bit-packed (as the level-geometry kernel hands it over): identical result
    words = torch.zeros(B, (S + 31) // 32, dtype=torch.int64)
    idx = torch.arange(S)
    for bi in range(B):
        words[bi].index_add_(0, idx // 32, keep[bi].long() << (idx % 32))
    bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).to(DEV)
    out_b, _ = k.ms_deform_attn_forward_fused(d[0], d[1], d[2], off, logits, d[4], False, d[5], keep_bits=bits)
    assert torch.equal(out_b, out)
    # a mask that is not 0 / 1 (255 for "real") packs like its != 0 image
    from egtr_amd.load_custom import pack_keep_bits
    assert torch.equal(pack_keep_bits((d[5].to(torch.uint8) * 255), B, S), bits)


@pytest.mark.parametrize("Lq", [50, 300, 1400])
def test_msda_fused_value_bias_in_kernel(Lq):
    """value_bias applied inside the kernel (times the sum of the in-range, unpadded corner weights) against sampling the
    finished values (W x + b, padded rows zeroed).  Lq <= 1024 takes the sample-split workgroup kernel, 1400 the
    wave-per-query one; 10 % of the samples fall outside the maps, 25 % of the tokens are padding."""
    k = _kernels()
    g = torch.Generator().manual_seed(90 + Lq)
    shapes = [(19, 32), (10, 16), (5, 8), (3, 4)]
    B, S = 2, sum(h * w for h, w in shapes)
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    raw = torch.randn(B, S, 8, 32, generator=g)
    bias = torch.randn(256, generator=g)
    off = torch.randn(B, Lq, 8, 4, 4, 2, generator=g) * 3
    logits = torch.randn(B, Lq, 8, 16, generator=g) * 2
    ref = torch.rand(B, Lq, 4, 2, generator=g) * 1.2 - 0.1
    keep = torch.rand(B, S, generator=g) > 0.25
    d = [t.to(DEV) for t in (raw, shp, lsi, off, logits, ref, keep, bias)]
    for km in (d[6], None):
        out, _ = k.ms_deform_attn_forward_fused(d[0], d[1], d[2], d[3], d[4], d[5], False, km, value_bias=d[7])
        full = d[0] + d[7].view(1, 1, 8, 32)
        if km is not None:
            full = torch.where(km[..., None, None], full, torch.zeros((), device=DEV))
        want, _ = k.ms_deform_attn_forward_fused(full.contiguous(), d[1], d[2], d[3], d[4], d[5], False, None)
        assert (out - want).abs().max().item() < 5e-6


# ------------------------------------------------------------------------------------------- bf16 epilogues
@pytest.mark.parametrize("rows", [1, 5, 200, 12537])
def test_add_layer_norm_bf16(rows):
    """egtr_add_layernorm_bf16 (bf16 storage, fp32 statistics) vs the fp64 LayerNorm of the bf16-rounded sum."""
    from egtr_amd.ops import add_layer_norm
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g).bfloat16()
    r = torch.randn(rows, 256, generator=g).bfloat16()
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g)); ln.bias.copy_(0.2 * torch.randn(256, generator=g))
    ln = ln.to(DEV).bfloat16()
    with torch.no_grad():
        y = add_layer_norm(x.to(DEV), r.to(DEV), ln)
    assert y.dtype == torch.bfloat16
    s_ = (x + r).double()   # the reference composition rounds the residual sum to bf16
    want = torch.nn.functional.layer_norm(s_, (256,), ln.weight.double().cpu(), ln.bias.double().cpu(), ln.eps)
    assert (y.cpu().double() - want).abs().max() < 2e-2 * max(1.0, float(want.abs().max()) / 4)


@pytest.mark.parametrize("rows,prow", [(8, 8), (5, 5), (603, 201), (12537, 12537), (2 * 4099, 4099)])
def test_add_layer_norm_pos_bf16(rows, prow):
    """egtr_add_layernorm_pos_bf16: the LayerNorm output is the one of egtr_add_layernorm_bf16 bit for bit (row counts that
    are not a multiple of the 8 rows of a workgroup included) and the second output is torch's bf16 `y + pos` of it."""
    from egtr_amd.ops import add_layer_norm, add_layer_norm_pos
    g = torch.Generator().manual_seed(rows + prow)
    x = torch.randn(rows, 256, generator=g).bfloat16().to(DEV)
    r = torch.randn(rows, 256, generator=g).bfloat16().to(DEV)
    pos = torch.randn(prow, 256, generator=g).bfloat16().to(DEV)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g)); ln.bias.copy_(0.2 * torch.randn(256, generator=g))
    ln = ln.to(DEV).bfloat16()
    with torch.no_grad():
        y0 = add_layer_norm(x, r, ln)
        y, yp = add_layer_norm_pos(x, r, ln, pos)
    assert y.dtype == torch.bfloat16 and yp.dtype == torch.bfloat16
    assert torch.equal(y, y0)
    assert torch.equal(yp, y0 + pos.repeat(rows // prow, 1))


@pytest.mark.parametrize("geom", ["small", "ragged"])
def test_level_geometry_bf16_rounds_like_the_reference_composition(geom):
    """egtr_level_geometry_bf16: position rows = bf16(bf16(sine) + level_embed), i.e. the reference's
    `position_embedding(..).to(bf16)` followed by its bf16 `+ level_embed[level]` (dd:2224, 2259) -- bit for bit against the
    fp32 kernel's sine part (level_embed = 0) pushed through exactly those two torch roundings; masks, valid ratios and
    reference points are the fp32 kernel's."""
    from egtr_amd import ops
    if geom == "small":
        B, H, W_, shapes = 3, 64, 80, [(8, 10), (4, 5), (2, 3), (1, 2)]
    else:
        B, H, W_, shapes = 3, 200, 333, [(25, 42), (13, 21), (7, 11), (4, 6)]
    pm = torch.ones(B, H, W_, dtype=torch.long)
    pm[1, H // 2:, :] = 0
    pm[2, :, W_ // 3:] = 0
    pm = pm.to(DEV)
    g = torch.Generator().manual_seed(5)
    le16 = torch.randn(4, 256, generator=g).bfloat16().to(DEV)
    two_pi = 2 * 3.141592653589793
    m0, sine, vr0, ref0, bits0 = ops.level_geometry(pm, shapes, torch.zeros(4, 256, device=DEV), 128, 10000, two_pi)
    m1, pos16, vr1, ref1, bits1 = ops.level_geometry(pm, shapes, le16, 128, 10000, two_pi)
    assert pos16.dtype == torch.bfloat16 and pos16.shape == sine.shape
    sizes = [h * w for h, w in shapes]
    per_tok = torch.cat([le16[l].view(1, 1, -1).expand(B, n, -1) for l, n in enumerate(sizes)], 1)
    assert torch.equal(pos16, sine.bfloat16() + per_tok)
    assert torch.equal(m0, m1) and torch.equal(vr0, vr1) and torch.equal(ref0, ref1)
    assert torch.equal(bits0, bits1)


def test_input_proj_groupnorm_flatten_bf16_matches_torch():
    """bf16 entry (bf16 convolution outputs in, bf16 tokens out, fp32 parameters and statistics) against the fp64 GroupNorm of
    the same bf16 inputs: one bf16 rounding of an O(1) output; odd plane sizes (2-byte aligned rows only)."""
    import torch.nn as nn
    from egtr_amd import ops
    torch.manual_seed(8)
    shapes = [(19, 33), (10, 17), (5, 9), (3, 3)]
    projs = nn.ModuleList([nn.Sequential(nn.Conv2d(8, 256, 1), nn.GroupNorm(32, 256)) for _ in shapes])
    with torch.no_grad():
        for p in projs:
            p[0].bias.normal_()
            p[1].weight.normal_()
            p[1].bias.normal_()
    projs = projs.bfloat16()
    xs = [(3.0 * torch.randn(2, 256, h, w) + 1.5).bfloat16() for h, w in shapes]
    with torch.no_grad():
        want = torch.cat([nn.functional.group_norm(x.double() + p[0].bias.double().view(1, -1, 1, 1), 32, p[1].weight.double(),
                                                   p[1].bias.double(), p[1].eps).flatten(2).transpose(1, 2)
                          for p, x in zip(projs, xs)], 1)
        got = ops.input_proj_groupnorm_flatten([x.to(DEV) for x in xs], projs.to(DEV))
        again = ops.input_proj_groupnorm_flatten([x.to(DEV) for x in xs], projs.to(DEV))   # cached fp32 parameters
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    err = (got.cpu().double() - want).abs()
    assert float((err / want.abs().clamp_min(1.0)).max()) < 2.0 ** -8     # half an ulp of bf16 (8 bits of mantissa)
    assert torch.equal(got, again)


def test_input_proj_groupnorm_tokens_bf16_matches_torch_and_the_nchw_kernel():
    """Token-major entry (channels-last backbone): [B, HW_l, 256] bf16 projections in, conv bias + GroupNorm(32) + concatenation
    out -- against the fp64 GroupNorm of the same bf16 inputs (half a bf16 ulp) and against the NCHW entry on the transposed
    inputs (the same fp32 arithmetic: equal up to the order of the statistics' sums, i.e. at most one bf16 ulp apart)."""
    import torch.nn as nn
    from egtr_amd import ops
    torch.manual_seed(9)
    shapes = [(19, 33), (10, 17), (5, 9), (3, 3)]
    projs = nn.ModuleList([nn.Sequential(nn.Conv2d(8, 256, 1), nn.GroupNorm(32, 256)) for _ in shapes])
    with torch.no_grad():
        for p in projs:
            p[0].bias.normal_()
            p[1].weight.normal_()
            p[1].bias.normal_()
    projs = projs.bfloat16()
    xs = [(3.0 * torch.randn(2, 256, h, w) + 1.5).bfloat16() for h, w in shapes]
    toks = [x.flatten(2).transpose(1, 2).contiguous().to(DEV) for x in xs]
    with torch.no_grad():
        want = torch.cat([nn.functional.group_norm(x.double() + p[0].bias.double().view(1, -1, 1, 1), 32, p[1].weight.double(),
                                                   p[1].bias.double(), p[1].eps).flatten(2).transpose(1, 2)
                          for p, x in zip(projs, xs)], 1)
        got = ops.input_proj_groupnorm_tokens(toks, projs.to(DEV))
        nchw = ops.input_proj_groupnorm_flatten([x.to(DEV) for x in xs], projs.to(DEV))
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    err = (got.cpu().double() - want).abs()
    assert float((err / want.abs().clamp_min(1.0)).max()) < 2.0 ** -8
    assert float(((got.float() - nchw.float()).abs() / nchw.float().abs().clamp_min(1.0)).max()) <= 2.0 ** -7
    # fp32 twin: against the fp64 GroupNorm and the NCHW fp32 entry
    projs32 = projs.float()
    xs32 = [x.float() for x in xs]
    with torch.no_grad():
        got32 = ops.input_proj_groupnorm_tokens([x.flatten(2).transpose(1, 2).contiguous().to(DEV) for x in xs32], projs32.to(DEV))
        nchw32 = ops.input_proj_groupnorm_flatten([x.to(DEV) for x in xs32], projs32.to(DEV))
    assert got32.dtype == torch.float32
    assert float((got32.cpu().double() - want).abs().max()) < 2e-5 and float((got32 - nchw32).abs().max()) < 2e-5


@pytest.mark.parametrize("rows,C", [(1, 8), (33, 64), (16 * 1000, 256), (977, 2048)])
def test_bias_act_rows_bf16_channels_last(rows, C):
    """egtr_bias_act_nhwc_bf16: y = act(x + bias[c] (+ residual)) on a [rows, C] bf16 matrix, in place, against torch in fp32
    rounded once."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).bfloat16().to(DEV)
    r = torch.randn(rows, C, generator=g).bfloat16().to(DEV)
    b = torch.randn(C, generator=g).to(DEV)
    for res, relu in ((None, True), (r, True), (r, False)):
        want = x.float() + b + (res.float() if res is not None else 0.0)
        want = (want.relu() if relu else want).bfloat16()
        got = ops.bias_act_rows_(x.clone(), b, res, relu)
        assert torch.equal(got, want), (res is not None, relu)
        if C % 4 == 0:   # fp32 twin (the same sums in the same order)
            x32, r32 = x.float(), (res.float() if res is not None else None)
            w32 = x32 + b + (r32 if r32 is not None else 0.0)
            assert torch.equal(ops.bias_act_rows_(x32.clone(), b, r32, relu), w32.relu() if relu else w32)


@pytest.mark.parametrize("N,C,H,Wd", [(2, 64, 25, 42), (1, 256, 100, 167), (3, 8, 3, 3), (1, 5, 1, 7), (2, 16, 50, 84)])
def test_bias_act_nchw_bf16(N, C, H, Wd):
    from egtr_amd.ops import bias_act_
    g = torch.Generator().manual_seed(N * C + H)
    x = torch.randn(N, C, H, Wd, generator=g).bfloat16()
    res = torch.randn(N, C, H, Wd, generator=g).bfloat16()
    b = torch.randn(C, generator=g)
    for use_res, relu in ((True, True), (False, True), (False, False)):
        want = x.float() + b.view(1, C, 1, 1) + (res.float() if use_res else 0)
        want = torch.relu(want) if relu else want
        y = bias_act_(x.clone().to(DEV), b.to(DEV), res.to(DEV) if use_res else None, relu=relu)
        assert y.dtype == torch.bfloat16
        assert torch.equal(y.cpu(), want.bfloat16()), (use_res, relu)


@pytest.mark.parametrize("shapes,B,Lq", [([(19, 32), (10, 16), (5, 8), (3, 4)], 2, None),
                                         ([(19, 32), (10, 16), (5, 8), (3, 4)], 3, 50), ([(9, 13), (5, 7)], 2, 33)])
def test_msda_fused_bf16_matches_fp32_composition(shapes, B, Lq):
    """egtr_msda_forward_fused_bf16 (bf16 storage, fp32 prologue + accumulation, optional padding mask) against the
    oracle evaluated on the SAME bf16-rounded operands."""
    k = _kernels()
    g = torch.Generator().manual_seed(17)
    L = len(shapes)
    P = 16 // L
    S = sum(h * w for h, w in shapes)
    Lq = Lq or S
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    bf = torch.bfloat16
    value = torch.randn(B, S, 8, 32, generator=g).to(bf)
    off = (3.0 * torch.randn(B, Lq, 8, L, P, 2, generator=g)).to(bf)
    logits = (2.0 * torch.randn(B, Lq, 8, L * P, generator=g)).to(bf)
    ref = (torch.rand(B, Lq, L, 2, generator=g) * 1.2 - 0.1).to(bf)
    keep = torch.rand(B, S, generator=g) > 0.2
    norm = torch.stack([shp[:, 1], shp[:, 0]], -1).float()
    loc = ref.float()[:, :, None, :, None, :] + off.float() / norm[None, None, None, :, None, :]
    attn = torch.softmax(logits.float(), -1).view(B, Lq, 8, L, P)
    for km in (None, keep):
        v = value.float() if km is None else value.float() * km[..., None, None]
        want = OM.msda_forward(v, shp, lsi, loc.contiguous(), attn.contiguous())
        got = k.ms_deform_attn_forward_fused_bf16(value.to(DEV), shp.to(DEV), lsi.to(DEV), off.to(DEV), logits.to(DEV),
                                                  ref.to(DEV), None if km is None else km.to(DEV))
        assert got.dtype == bf
        assert (got.cpu().float() - want).abs().max() < 2e-2 * max(1.0, float(want.abs().max()))



# ------------------------------------------------------------------------------------------------ matcher
def test_device_assignment_equals_scipy_on_random_and_tied_matrices():
    """egtr_hungarian_match_f32 in solve-only mode against scipy.optimize.linear_sum_assignment, index for index, on
    1000+ cost matrices: continuous, integer-valued with many exact ties, constant, T from 1 to beyond N (wide problems
    are not transposed), batches of ragged sizes."""
    from scipy.optimize import linear_sum_assignment as sp
    from egtr_amd.ops import hungarian_match
    rng = np.random.default_rng(5)
    n_checked = 0
    for N in (200, 37, 300):
        for rep in range(14):
            B = 24
            sizes = [int(rng.integers(1, 63)) for _ in range(B)]
            if rep == 0:
                sizes[:4] = [1, N, N + 3, min(N + 40, 330)]
            mats = []
            for i, T in enumerate(sizes):
                kind = (rep + i) % 5
                c = rng.standard_normal((N, T)).astype(np.float32)
                if kind == 1:
                    c = np.round(c)                                  # integer costs: exact ties everywhere
                elif kind == 2:
                    c = rng.integers(0, 2, (N, T)).astype(np.float32)
                elif kind == 3:
                    c = np.full((N, T), 0.25, dtype=np.float32)      # constant (scipy issue 11602 ordering)
                elif kind == 4:
                    c = np.round(c * 4) / 4
                mats.append(c)
            pi, ti, mc, n_out, status = hungarian_match(None, None, None, 1, 1, 1,
                                                        cost_in=[torch.from_numpy(m).to(DEV) for m in mats],
                                                        want_status=True)
            pi, ti, mc = pi.cpu().numpy(), ti.cpu().numpy(), mc.cpu().numpy()
            assert status.cpu().abs().sum() == 0
            o = 0
            for m, n in zip(mats, n_out):
                a, b = sp(m)
                assert n == len(a)
                assert np.array_equal(pi[o:o + n], a) and np.array_equal(ti[o:o + n], b), (N, rep, m.shape)
                assert np.array_equal(mc[o:o + n], m[a, b])
                o += n
                n_checked += 1
    assert n_checked >= 1000
    # an image without targets between two ordinary ones: empty assignment, neighbours unaffected
    mats = [rng.standard_normal((50, 7)).astype(np.float32), np.zeros((50, 0), dtype=np.float32),
            rng.standard_normal((50, 3)).astype(np.float32)]
    pi, ti, mc, n_out = hungarian_match(None, None, None, 1, 1, 1, cost_in=[torch.from_numpy(m).to(DEV) for m in mats])
    assert n_out == [7, 0, 3]
    a, b = sp(mats[2])
    assert np.array_equal(pi[7:].cpu().numpy(), a) and np.array_equal(ti[7:].cpu().numpy(), b)
    # invalid entries: scipy raises, the kernel flags the image and returns -1 indices
    bad = np.zeros((8, 3), dtype=np.float32)
    bad[2, 1] = np.nan
    pi, ti, mc, n_out, status = hungarian_match(None, None, None, 1, 1, 1, cost_in=[torch.from_numpy(bad).to(DEV)],
                                                want_status=True)
    assert int(status[0]) == 1 and (pi.cpu() == -1).all()


def test_device_matcher_cost_matrix_and_indices_vs_reference_composition(golden_dir):
    """The fused cost + assignment launch against (a) the reference's tensor composition of the cost matrix evaluated by
    PyTorch on the same device and on the CPU (1e-5; the kernel follows the reference's fp32 operation order), (b) scipy on
    that matrix (indices identical), for VG-sized heads, ragged target counts, with and without the smoothing offset."""
    from scipy.optimize import linear_sum_assignment as sp
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    from egtr_amd.ops import hungarian_match
    g = torch.Generator().manual_seed(3)
    B, N, K = 5, 200, 151
    logits = torch.randn(B, N, K, generator=g) * 3
    cxcy = torch.rand(B, N, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(B, N, 2, generator=g) * 0.3 + 0.02
    boxes = torch.cat([cxcy, wh], -1)
    targets = []
    for T in (30, 1, 17, 62, 5):
        targets.append({"class_labels": torch.randint(0, K - 1, (T,), generator=g),
                        "boxes": torch.cat([torch.rand(T, 2, generator=g) * 0.6 + 0.2,
                                            torch.rand(T, 2, generator=g) * 0.3 + 0.02], -1)})
    targets[2]["boxes"][:3] = boxes[2, :3]            # exact box matches (zero L1, GIoU = 1)
    for smoothing in (0.0, 1e-14):
        m = DeformableDetrHungarianMatcher(class_cost=2.0, bbox_cost=5.0, giou_cost=2.0, smoothing=smoothing)
        # reference composition on the CPU (the reference's own path) -> cost blocks + scipy indices
        kind, cm, sizes, _ = m.prepare({"logits": logits, "pred_boxes": boxes}, targets)
        assert kind == "host"
        idx_ref, cost_ref = m.finish((kind, cm, sizes, torch.device("cpu")))
        if smoothing:
            cmn, iss = m._smoothing_scalars()
            cm = cm - cmn + iss
        dt = [{k: v.to(DEV) for k, v in t.items()} for t in targets]
        cmn, iss = (m._smoothing_scalars() if smoothing else (None, None))
        pi, ti, mc, n_out, blocks = hungarian_match(logits.to(DEV), boxes.to(DEV), dt, 2.0, 5.0, 2.0,
                                                    float(cmn) if smoothing else None, float(iss) if smoothing else None,
                                                    want_cost=True)
        o = 0
        for i, (blk, n) in enumerate(zip(blocks, n_out)):
            want = cm[i].split(sizes, -1)[i]
            assert (blk.cpu() - want).abs().max() < 2e-5 * max(1.0, float(want.abs().max())), (i, smoothing)
            a, b = sp(blk.cpu().numpy())               # scipy on the DEVICE's matrix: identical indices
            assert np.array_equal(pi[o:o + n].cpu().numpy(), a) and np.array_equal(ti[o:o + n].cpu().numpy(), b)
            # and the reference's own result (CPU composition + scipy): identical unless a near-tie flips
            assert np.array_equal(a, idx_ref[i][0].numpy()) and np.array_equal(b, idx_ref[i][1].numpy()), i
            assert (mc[o:o + n].cpu() - cost_ref[i]).abs().max() < 2e-5 * max(1.0, float(cost_ref[i].abs().max()))
            o += n
        # the module path on device tensors returns the same thing as device tensors
        idx_dev, cost_dev = m({"logits": logits.to(DEV), "pred_boxes": boxes.to(DEV)}, dt)
        for (a, b), (ra, rb) in zip(idx_dev, idx_ref):
            assert a.is_cuda and np.array_equal(a.cpu().numpy(), ra.numpy()) and np.array_equal(b.cpu().numpy(), rb.numpy())


# ------------------------------------------------------------------------------------------- relation loss
def _criterion(N, R, smoothing=1e-14):
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    from egtr_amd.egtr import SceneGraphGenerationLoss
    m = DeformableDetrHungarianMatcher(class_cost=2.0, bbox_cost=5.0, giou_cost=2.0, smoothing=smoothing)
    return SceneGraphGenerationLoss(
        matcher=m, num_object_queries=N, num_classes=12, num_rel_labels=R, eos_coef=0.1,
        losses=["relations"], smoothing=smoothing, rel_sample_negatives=80, rel_sample_nonmatching=80,
        model_training=True, focal_alpha=0.25, rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True)


@pytest.mark.parametrize("B,N,R,Ts,nrel", [
    (3, 40, 7, (5, 12, 1), 3),        # small: k1 limited by the number of false candidates
    (2, 200, 50, (30, 9), 3),         # VG-sized
    (2, 64, 9, (6, 0), 2),            # an image without targets
    (1, 200, 50, (17,), 0),           # no relation at all: the reference's mean of an empty tensor (NaN), zero gradients
])
def test_relation_loss_kernel_vs_reference_loop(B, N, R, Ts, nrel):
    """egtr_relation_loss_f32 (value + gradient) against the line-by-line mirror of the reference's loss_relations /
    _loss_relations (egtr:754-923: permutation, nonzero() index lists, topk over gathered scores, BCE) evaluated by
    PyTorch on the CPU in float64."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(17 + N)
    crit = _criterion(N, R)
    pred_rel = (torch.randn(B, N, N, R, generator=g) * 2)
    pred_conn = torch.randn(B, N, N, 1, generator=g)
    targets, indices, costs = [], [], []
    for T in Ts:
        rel = torch.zeros(N, N, R)
        if T > 0:
            so = torch.randint(0, T, (nrel * T, 2), generator=g)
            rr = torch.randint(0, R, (nrel * T,), generator=g)
            keep = so[:, 0] != so[:, 1]
            rel[so[keep, 0], so[keep, 1], rr[keep]] = 1.0
        targets.append({"rel": rel})
        src = torch.randperm(N, generator=g)[:T].sort()[0]
        tgt = torch.randperm(T, generator=g)
        indices.append((src, tgt))
        costs.append(torch.randn(T, generator=g) * 3)
    # reference mirror on the CPU, float64
    pr64 = pred_rel.double().requires_grad_(True)
    pc64 = pred_conn.double().requires_grad_(True)
    crit.nonmatching_cost = crit.nonmatching_cost.double()
    want = crit.loss_relations({"pred_rel": pr64, "pred_connectivity": pc64},
                               [{"rel": t["rel"].double()} for t in targets], indices, [c.double() for c in costs], 1.0)
    nan_case = bool(torch.isnan(want["loss_rel"]))
    if not nan_case:
        (want["loss_rel"] * 1.5 + want["loss_connectivity"] * 0.5).backward()
    else:
        (want["loss_connectivity"] * 0.5).backward()
    # device kernel
    prd = pred_rel.to(DEV).requires_grad_(True)
    pcd = pred_conn.to(DEV).requires_grad_(True)
    l_rel, l_conn = ops.relation_losses(prd, pcd, [{"rel": t["rel"].to(DEV)} for t in targets],
                                        [(a.to(DEV), b_.to(DEV)) for a, b_ in indices], [c.to(DEV) for c in costs],
                                        float(crit.nonmatching_cost), 80, 80)
    assert abs(float(l_conn) - float(want["loss_connectivity"])) < 1e-5 * max(1.0, abs(float(want["loss_connectivity"])))
    if nan_case:
        assert bool(torch.isnan(l_rel))
        (l_conn * 0.5).backward()
        assert prd.grad is None or float(prd.grad.abs().max()) == 0.0
    else:
        assert abs(float(l_rel) - float(want["loss_rel"])) < 1e-5 * max(1.0, abs(float(want["loss_rel"])))
        (l_rel * 1.5 + l_conn * 0.5).backward()
        assert (prd.grad.cpu().double() - pr64.grad).abs().max() < 1e-7
    assert (pcd.grad.cpu().double() - pc64.grad).abs().max() < 1e-7


# --------------------------------------------------------------------------------------- boundary dtypes (fp64, bf16 bwd)
@pytest.mark.parametrize("case", ["a", "b"])
def test_msda_float64_entries_vs_reference_golden(golden_dir, case):
    """egtr_msda_forward_f64 / _backward_f64 through the module boundary (autograd Function) against the reference's own
    float64 outputs and autograd gradients (tests/golden/msda.npz), 1e-12: the reference extension dispatches double
    (ms_deform_attn_cuda.cu:67, 137)."""
    import json
    from egtr_amd.ops import MultiScaleDeformableAttentionFunction as F
    g = np.load(f"{golden_dir}/msda.npz")
    c = json.loads(str(g[f"{case}_case"]))
    x = W.make_msda_inputs(c["seed"], c["B"], c["Lq"], c["M"], c["D"], [tuple(s) for s in c["shapes"]], c["P"],
                           dtype=torch.float64)
    v = x["value"].to(DEV).requires_grad_(True)
    loc = x["loc"].to(DEV).requires_grad_(True)
    at = x["attn"].to(DEV).requires_grad_(True)
    out = F.apply(v, x["shapes"].to(DEV), x["lsi"].to(DEV), loc, at, 64)
    assert out.dtype == torch.float64
    out.backward(x["grad_out"].to(DEV))
    assert (out.detach().cpu() - torch.from_numpy(g[f"{case}_f64_out"])).abs().max() < 1e-12
    assert (v.grad.cpu() - torch.from_numpy(g[f"{case}_f64_grad_value"])).abs().max() < 1e-11
    assert (at.grad.cpu() - torch.from_numpy(g[f"{case}_f64_grad_attn"])).abs().max() < 1e-11
    assert (loc.grad.cpu() - torch.from_numpy(g[f"{case}_f64_grad_loc"])).abs().max() < 2e-10


@pytest.mark.parametrize("B,Lq,shapes", [(2, 300, [(19, 32), (10, 16), (5, 8), (3, 4)]),
                                         (1, 820, [(19, 32), (10, 16), (5, 8), (3, 4)])])    # decoder- and encoder-shaped
def test_msda_bf16_backward(B, Lq, shapes):
    """bf16 training (stress configuration): value / upstream gradient in bf16, geometry and gradients in fp32 -- against
    the fp32 oracle evaluated on the bf16-rounded operands: grad_loc / grad_attn 1e-3 relative (fp32 arithmetic on the
    same inputs), grad_value rounded to bf16 (2^-8 relative)."""
    from egtr_amd.ops import MultiScaleDeformableAttentionFunction as F
    x = W.make_msda_inputs(77, B, Lq, 8, 32, shapes, 4)
    vb = x["value"].to(torch.bfloat16)
    gb = x["grad_out"].to(torch.bfloat16)
    rgv, rgl, rga = OM.msda_backward(vb.float(), x["shapes"], x["lsi"], x["loc"], x["attn"], gb.float())
    v = vb.to(DEV).requires_grad_(True)
    loc = x["loc"].to(DEV).requires_grad_(True)
    at = x["attn"].to(DEV).requires_grad_(True)
    out = F.apply(v, x["shapes"].to(DEV), x["lsi"].to(DEV), loc, at, 64)
    assert out.dtype == torch.bfloat16
    out.backward(gb.to(DEV))
    assert v.grad.dtype == torch.bfloat16 and loc.grad.dtype == torch.float32
    assert (v.grad.float().cpu() - rgv).abs().max() < 2 ** -7 * max(1.0, float(rgv.abs().max()))
    assert (at.grad.cpu() - rga).abs().max() < 1e-3 * max(1.0, float(rga.abs().max()))
    assert (loc.grad.cpu() - rgl).abs().max() < 1e-3 * max(1.0, float(rgl.abs().max()))


@pytest.mark.parametrize("B,Lq,shapes", [(2, 0, [(19, 32), (10, 16), (5, 8), (3, 4)]), (2, 200, [(19, 32), (10, 16), (5, 8), (3, 4)]),
                                         (1, 5000, [(19, 32), (10, 16), (5, 8), (3, 4)])])
def test_msda_bf16_backward_reads_bf16_operands_directly(B, Lq, shapes):
    """egtr_msda_backward_bf16 on the model's shapes needs no workspace (the kernels widen the bf16 operands on load) and gives
    the fp32 kernels' results on the widened operands: grad_loc / grad_attn bit for bit (same instructions behind the load);
    grad_value to the order of its float atomics.  Lq = 0: encoder-shaped (Lq = S: wave-per-query kernel + value-tile
    kernel); 200: the decoder's four-waves-per-query kernel; 5000: one wave per query with atomics."""
    k = _kernels()
    from egtr_amd import _lib
    S = sum(h * w for h, w in shapes)
    x = W.make_msda_inputs(91, B, Lq or S, 8, 32, shapes, 4)
    assert int(_lib.lib().egtr_msda_backward_bf16_workspace_floats(B, S, 8, 32, 4, Lq or S, 4)) == 0
    vb, gb = x["value"].to(torch.bfloat16).to(DEV), x["grad_out"].to(torch.bfloat16).to(DEV)
    shp, lsi, loc, at = x["shapes"].to(DEV), x["lsi"].to(DEV), x["loc"].to(DEV), x["attn"].to(DEV)
    gv, gl, ga = k.ms_deform_attn_backward(vb, shp, lsi, loc, at, gb, 64)
    rv, rl, ra = k.ms_deform_attn_backward(vb.float(), shp, lsi, loc, at, gb.float(), 64)
    assert gv.dtype == torch.bfloat16 and torch.equal(gl, rl) and torch.equal(ga, ra)
    assert (gv.float() - rv).abs().max() <= 2 ** -7 * max(1.0, float(rv.abs().max()))


@pytest.mark.parametrize("B,N,C,Ts", [(2, 200, 150, (30, 9)), (3, 40, 12, (5, 12, 1)), (2, 64, 601, (6, 0)),
                                      (1, 300, 150, (40,))])
def test_detection_loss_kernel_vs_tensor_composition(B, N, C, Ts):
    """labels / cardinality / boxes of one output set in one launch (egtr_detection_loss_f32) against the line-by-line
    tensor composition (egtr:611-712) with autograd: values and gradients w.r.t. logits and boxes, incl. an image
    without targets and the aux-style weighting of the three terms."""
    from egtr_amd import ops
    from egtr_amd.deformable_detr import DeformableDetrHungarianMatcher
    from egtr_amd.egtr import SceneGraphGenerationLoss
    rng = W.rng_inputs(2300 + N + C)
    m = DeformableDetrHungarianMatcher(class_cost=2.0, bbox_cost=5.0, giou_cost=2.0, smoothing=1e-14)
    crit = SceneGraphGenerationLoss(
        matcher=m, num_object_queries=N, num_classes=C, num_rel_labels=7, eos_coef=0.1,
        losses=["labels", "cardinality", "boxes"], smoothing=1e-14, rel_sample_negatives=80, rel_sample_nonmatching=80,
        model_training=True, focal_alpha=0.25, rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True)
    logits = torch.from_numpy(rng.standard_normal((B, N, C)) * 2).float().to(DEV)
    cxcy = rng.uniform(0.2, 0.8, (B, N, 2))
    wh = rng.uniform(0.05, 0.4, (B, N, 2))
    boxes = torch.from_numpy(np.concatenate([cxcy, wh], -1)).float().to(DEV)
    targets = []
    for t in Ts:
        tc = rng.uniform(0.2, 0.8, (t, 2))
        tw = rng.uniform(0.05, 0.4, (t, 2))
        targets.append({"class_labels": torch.from_numpy(rng.integers(0, C, t)).long().to(DEV),
                        "boxes": torch.from_numpy(np.concatenate([tc, tw], -1)).float().to(DEV)})
    weights = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0}

    def run(fused):
        lg = logits.clone().requires_grad_(True)
        bx = boxes.clone().requires_grad_(True)
        out = {"logits": lg, "pred_boxes": bx}
        indices, costs = m(out, targets)
        assert getattr(indices, "flat", None) is not None
        if not fused:
            indices = list(indices)      # plain list: the tensor composition
        losses = crit(out, targets, matched=(indices, costs))
        total = sum(losses[k] * w for k, w in weights.items())
        total.backward()
        return {k: float(v) for k, v in losses.items()}, lg.grad.clone(), bx.grad.clone()

    lf, glf, gbf = run(True)
    lr, glr, gbr = run(False)
    assert set(lf) == set(lr) == {"loss_ce", "loss_bbox", "loss_giou", "cardinality_error"}
    for k in lr:
        assert abs(lf[k] - lr[k]) < 2e-5 * max(1.0, abs(lr[k])), (k, lf[k], lr[k])
    assert (glf - glr).abs().max() < 2e-5 * max(1e-3, float(glr.abs().max()))
    assert (gbf - gbr).abs().max() < 2e-5 * max(1e-3, float(gbr.abs().max()))


def test_clamp_nonfinite_matches_reference_branch():
    """dd:1346-1351: untouched (same storage) when every element is finite; clamped to +-(finfo.max - 1000) when an inf or
    a NaN is present (NaN stays NaN); gradients pass except through clamped / NaN elements."""
    from egtr_amd import ops
    cv = torch.finfo(torch.float32).max - 1000
    x = torch.randn(1000, 256, device=DEV)
    xin = x.clone().requires_grad_(True)
    y = ops.clamp_nonfinite_(xin * 1.0)
    assert torch.equal(y, x)
    y.sum().backward()
    assert torch.equal(xin.grad, torch.ones_like(x))
    x2 = x.clone()
    x2[3, 7] = float("inf")
    x2[5, 0] = float("-inf")
    x2[9, 9] = float("nan")
    x2[11, 1] = 3e38   # finite, inside the clamp range: untouched
    x2in = x2.clone().requires_grad_(True)
    y2 = ops.clamp_nonfinite_(x2in * 1.0)
    ref = torch.clamp(x2, min=-cv, max=cv)
    assert torch.equal(torch.nan_to_num(y2.detach(), nan=123.0), torch.nan_to_num(ref, nan=123.0))
    assert y2[3, 7] == cv and y2[5, 0] == -cv and torch.isnan(y2[9, 9]) and y2[11, 1] == 3e38
    y2.nan_to_num(0.0).sum().backward()
    g = xin.grad * 0 + x2in.grad
    assert g[3, 7] == 0 and g[5, 0] == 0 and g[9, 9] == 0 and g[11, 1] == 1 and g[0, 0] == 1
    # odd sizes / tails
    z = torch.randn(4 * 77 + 3, device=DEV)
    z[-1] = float("inf")
    assert ops.clamp_nonfinite_(z.clone())[-1] == cv


@pytest.mark.parametrize("M,K,N,relu", [(12537, 256, 1024, True), (12537, 1024, 256, False), (5000, 256, 384, False),
                                        (4133, 256, 128, False), (4200, 96, 200, True), (4200, 64, 150, False)])
def test_token_linear_training_function_matches_float64(M, K, N, relu):
    """Token-sized nn.Linear (+ ReLU) in training (ops.TokenLinearFunction: split-bf16 GEMM for the forward and the data
    gradient where the shape allows, vendor GEMM otherwise and for the weight gradient, egtr_column_sum_f32 for the bias
    gradient / ReLU mask): output and all three gradients against float64, no further from it than 2.5x plain fp32
    autograd (+ a floor).  The ReLU mask of the reference is taken from the output under test (entries within rounding
    of zero may fall on either side)."""
    from egtr_amd import ops
    rng = W.rng_inputs(3100 + N)
    x = torch.from_numpy(rng.standard_normal((M, K))).float().to(DEV)
    w = torch.from_numpy(rng.standard_normal((N, K)) / np.sqrt(K)).float().to(DEV)
    b = torch.from_numpy(rng.standard_normal(N) * 0.1).float().to(DEV)
    go = torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV)
    outs = []
    for fn in ("custom", "autograd"):
        xi, wi, bi = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if fn == "custom":
            y = ops.linear(xi, wi, bi, 1.0, relu)
            assert y.grad_fn.name().startswith("TokenLinearFunction")
        else:
            y = torch.nn.functional.linear(xi, wi, bi)
            y = torch.relu(y) if relu else y
        (y * go).sum().backward()
        outs.append((y.detach(), xi.grad, wi.grad, bi.grad))
    y64 = x.double() @ w.double().t() + b.double()
    for k, (yk, gx, gw, gb) in enumerate(outs):
        mask = (yk > 0).double() if relu else torch.ones_like(y64)
        g64 = go.double() * mask
        refs = (torch.relu(y64) if relu else y64, g64 @ w.double(), g64.t() @ x.double(), g64.sum(0))
        errs = [float((a.double() - r).abs().max() / max(1.0, float(r.abs().max()))) for a, r in zip((yk, gx, gw, gb), refs)]
        outs[k] = errs
    for e_custom, e_plain in zip(*outs):
        assert e_custom <= 2.5 * e_plain + 2e-6, (outs[0], outs[1])


def test_gemm_split_tile_kernel_equals_host_tiling():
    """egtr_gemm_split_tile_weights_f32 (one launch per weight per training step) is bit-identical to the torch
    composition ops.gemm_split_weights, for W and for W^T read in place, incl. a strided source and denormal / huge
    entries."""
    from egtr_amd import ops
    rng = W.rng_inputs(3200)
    for N, K in ((256, 256), (1024, 256), (128, 1024), (384, 32)):
        big = torch.from_numpy(rng.standard_normal((N, K + 8))).float().to(DEV)
        big[0, 0], big[1, 1], big[2, 2], big[3, 3] = 1e-40, -3e38, 0.0, 1.0 + 2.0 ** -9
        w = big[:, :K]                                   # row stride K + 8
        assert torch.equal(ops.gemm_split_tile(w).view(torch.int16), ops.gemm_split_weights(w).view(torch.int16))
        if K % 128 == 0 and N % 32 == 0:
            wt = ops.gemm_split_tile(w, transposed=True)     # tiles of W^T [K, N]
            ref_t = ops.gemm_split_weights(w.t().contiguous()).view(torch.int16)
            assert torch.equal(wt.view(torch.int16), ref_t)
            if N % 128 == 0:                                 # both in one launch
                a, b = ops.gemm_split_tile_pair(w)
                assert torch.equal(a.view(torch.int16), ops.gemm_split_weights(w).view(torch.int16))
                assert torch.equal(b.view(torch.int16), ref_t)


@pytest.mark.parametrize("rows", [1, 7, 800, 16385, 50148])
def test_add_layernorm_backward_kernel_vs_float64(rows):
    """egtr_add_layernorm_backward_f32 (gradient of x + residual, gamma, beta in one pass) against float64 autograd, no
    further from it than 3x fp32 autograd (+ floor); bit-reproducible."""
    from egtr_amd import ops
    rng = W.rng_inputs(3300 + rows)
    x = torch.from_numpy(rng.standard_normal((rows, 256))).float().to(DEV)
    r = torch.from_numpy(rng.standard_normal((rows, 256)) * 2 + 0.5).float().to(DEV)
    gy = torch.from_numpy(rng.standard_normal((rows, 256))).float().to(DEV)
    ln = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(rng.standard_normal(256)).float() * 0.3 + 1.0)
        ln.bias.copy_(torch.from_numpy(rng.standard_normal(256)).float() * 0.1)

    def run(kind):
        dt = torch.float64 if kind == "f64" else torch.float32
        xi, ri = x.to(dt).requires_grad_(True), r.to(dt).requires_grad_(True)
        wi, bi = ln.weight.detach().to(dt).requires_grad_(True), ln.bias.detach().to(dt).requires_grad_(True)
        if kind == "hip":
            y = ops.AddLayerNormFunction.apply(xi, ri, wi, bi, ln.eps)
        else:
            y = torch.nn.functional.layer_norm(xi + ri, (256,), wi, bi, ln.eps)
        y.backward(gy.to(dt))
        return [t.grad for t in (xi, ri, wi, bi)]

    ref, hip, plain = run("f64"), run("hip"), run("f32")
    assert torch.equal(hip[0], hip[1])
    for a, p, t in zip(hip, plain, ref):
        scale = max(1.0, float(t.abs().max()))
        assert float((a.double() - t).abs().max()) / scale <= 3 * float((p.double() - t).abs().max()) / scale + 2e-6
    again = run("hip")
    assert all(torch.equal(a, b) for a, b in zip(hip, again))


@pytest.mark.parametrize("M,N", [(800, 256), (1, 4), (2048, 151), (2049, 151), (4133, 128), (800, 1024)])
def test_column_sum_kernel_and_relu_mask(M, N):
    """egtr_column_sum_f32: the single-launch path (M <= 2048), the scalar-column path (N % 4 != 0) and the masked variant
    (g * [y > 0] written out) against float64."""
    from egtr_amd import ops
    rng = W.rng_inputs(3400 + M + N)
    g = torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV)
    y = torch.relu(torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV))
    s = ops.column_sum(g)
    ref = g.double().sum(0)
    assert (s.double() - ref).abs().max() <= 1e-5 * max(1.0, float(g.abs().sum(0).max()))
    gm, sm = ops.column_sum(g, relu_output=y)
    assert torch.equal(gm, g * (y > 0))
    refm = gm.double().sum(0)
    assert (sm.double() - refm).abs().max() <= 1e-5 * max(1.0, float(g.abs().sum(0).max()))
    assert torch.equal(ops.column_sum(g), s)
    g2 = g.clone()
    gm2, sm2 = ops.column_sum(g2, relu_output=y, inplace=True)   # masked gradient written over g
    assert gm2.data_ptr() == g2.data_ptr() and torch.equal(gm2, gm) and torch.equal(sm2, sm)


@pytest.mark.parametrize("B,Lq,ref_dim,ref_grad", [(2, 700, 2, False), (1, 33, 2, True), (2, 50, 4, True), (4, 12537, 2, False)])
def test_msda_geometry_function_matches_reference_composition(B, Lq, ref_dim, ref_grad):
    """ops.MSDAGeometryFunction (softmax + sampling locations of dd:1055-1073 and their backward, one HIP pass each)
    against the reference's ATen composition: forward to fp32 rounding, gradients against float64 autograd no further
    than 3x the fp32 composition (+ floor).  The offsets arrive as a column block of a wider buffer (row stride 384)."""
    from egtr_amd import ops
    M, L, P = 8, 4, 4
    rng = W.rng_inputs(3500 + Lq)
    both = torch.from_numpy(rng.standard_normal((B, Lq, 384)) * 2).float().to(DEV)
    ref = torch.from_numpy(rng.uniform(0.1, 0.9, (B, Lq, L, ref_dim))).float().to(DEV)
    shapes = torch.tensor([[75, 125], [38, 63], [19, 32], [10, 16]], dtype=torch.int64, device=DEV)
    g_loc = torch.from_numpy(rng.standard_normal((B, Lq, M, L, P, 2))).float().to(DEV)
    g_att = torch.from_numpy(rng.standard_normal((B, Lq, M, L, P))).float().to(DEV)

    def run(kind):
        dt = torch.float64 if kind == "f64" else torch.float32
        bi = both.to(dt).requires_grad_(True)
        ri = ref.to(dt).requires_grad_(ref_grad)
        off, lg = bi[..., :256], bi[..., 256:]
        if kind == "hip":
            loc, att = ops.MSDAGeometryFunction.apply(off, lg, ri, shapes, M, L, P)
        elif kind == "hip_both":     # one Linear produced both blocks: gradient written as one buffer
            loc, att = ops.MSDAGeometryFunction.apply(bi, None, ri, shapes, M, L, P)
        else:
            att = torch.softmax(lg.reshape(B, Lq, M, L * P), -1).view(B, Lq, M, L, P)
            o6 = off.reshape(B, Lq, M, L, P, 2)
            if ref_dim == 2:
                norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
                loc = ri[:, :, None, :, None, :] + o6 / norm[None, None, None, :, None, :]
            else:
                loc = ri[:, :, None, :, None, :2] + o6 / P * ri[:, :, None, :, None, 2:] * 0.5
        ((loc * g_loc.to(dt)).sum() + (att * g_att.to(dt)).sum()).backward()
        return loc.detach(), att.detach(), bi.grad, (ri.grad if ref_grad else None)

    assert ops.msda_geometry_supported(both[..., :256], both[..., 256:], ref, M, L, P)
    t, h, p = run("f64"), run("hip"), run("f32")
    hb = run("hip_both")
    assert all(x is y or torch.equal(x, y) for x, y in zip(h, hb))
    assert (h[0] - p[0]).abs().max() <= 1e-6 and (h[1] - p[1]).abs().max() <= 1e-6
    for a, b, r in zip(h, p, t):
        if r is None:
            continue
        scale = max(1.0, float(r.abs().max()))
        assert float((a.double() - r).abs().max()) / scale <= 3 * float((b.double() - r).abs().max()) / scale + 2e-6


@pytest.mark.parametrize("M,N,K", [(50148, 256, 256), (12537, 384, 256), (4100, 128, 1024), (33, 128, 128), (777, 256, 128)])
def test_linear_split_bf16_wgrad_is_fp32_accurate(M, N, K):
    """egtr_linear_split_bf16_wgrad_f32 (g^T x over the token rows, split-K with a fixed-order reduction) against float64:
    within 2.5x of the vendor fp32 GEMM's own error (+ floor), strided operands, ragged last chunk, bit-reproducible."""
    from egtr_amd import ops
    rng = W.rng_inputs(3600 + M)
    gbig = torch.from_numpy(rng.standard_normal((M, N + 128))).float().to(DEV)
    g = gbig[:, 128:]                                   # column block of a wider buffer
    x = torch.from_numpy(rng.standard_normal((M, K)) + 0.3).float().to(DEV)
    ref = g.double().t() @ x.double()
    out = ops.linear_split_bf16_wgrad(g, x)
    vend = g.t() @ x
    scale = float(ref.abs().max())
    e_out, e_vend = float((out.double() - ref).abs().max()) / scale, float((vend.double() - ref).abs().max()) / scale
    assert e_out <= 2.5 * e_vend + 1e-6, (e_out, e_vend)
    assert torch.equal(out, ops.linear_split_bf16_wgrad(g, x))


@pytest.mark.parametrize("M,N", [(160000, 256), (777, 128), (65, 1024)])
def test_weighted_column_sum_vs_float64(M, N):
    """egtr_weighted_column_sum_f32 ([1, M] x [M, N]) against float64, bit-reproducible."""
    from egtr_amd import ops
    rng = W.rng_inputs(3800 + M)
    g = torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV)
    w = torch.from_numpy(rng.standard_normal((M, 1))).float().to(DEV)
    out = ops.weighted_column_sum(g, w)
    ref = (w.double().t() @ g.double()).reshape(-1)
    assert (out.double() - ref).abs().max() <= 1e-5 * float((w.abs() * g.abs()).sum(0).max())
    assert torch.equal(out, ops.weighted_column_sum(g, w))


@pytest.mark.parametrize("B,N", [(1, 200), (2, 37), (1, 16)])
def test_linear_grouped_layernorm_prologue(B, N):
    """egtr_linear_grouped_ln_f32: groups whose input is a DeferredLayerNorm evaluate LayerNorm(a + b) [+ pos] themselves
    (dd:1437-1438, 1456-1457, 1466-1468 folded into the consuming nn.Linear), exactly one of them stores the LayerNorm
    result; plain groups in the same launch are unaffected.  Against the fp32 torch composition, 2e-5."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(5 + N)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)  # noqa: E731
    a, b, pos, other = r(B, N, 256), r(B, N, 256, sc=2.0), r(1, N, 256), r(B, N, 256)
    ln = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * r(256))
        ln.bias.copy_(0.2 * r(256))
    ws = [r(n, 256, sc=1 / 16) for n in (256, 256, 384, 1024, 256)]
    bs = [r(n, sc=0.1) for n in (256, 256, 384, 1024, 256)]
    with torch.no_grad():
        y = ln(a + b)
        want = [torch.nn.functional.linear(y + pos, ws[0], bs[0]) * 0.25, torch.nn.functional.linear(y + pos, ws[1], bs[1]),
                torch.nn.functional.linear(y, ws[2], bs[2]), torch.relu(torch.nn.functional.linear(y, ws[3], bs[3])),
                torch.nn.functional.linear(other, ws[4], bs[4])]
        d = ops.DeferredLayerNorm(a, b, ln)
        got = ops.linear_grouped([dict(x=d, pos=pos[0], w=ws[0], b=bs[0], alpha=0.25), dict(x=d, pos=pos[0], w=ws[1], b=bs[1]),
                                  dict(x=d, w=ws[2], b=bs[2]), dict(x=d, w=ws[3], b=bs[3], relu=True),
                                  dict(x=other, w=ws[4], b=bs[4])])
    assert d.done and (d.out - y).abs().max() < 2e-5
    for gt, wt in zip(got, want):
        assert gt.shape == wt.shape and (gt - wt).abs().max() < 2e-5 * max(1.0, float(wt.abs().max()))
    # a consumed DeferredLayerNorm is a plain tensor for later launches; an unconsumed one can be materialised directly
    with torch.no_grad():
        again = ops.linear_grouped([dict(x=d, w=ws[2], b=bs[2])])[0]
        d2 = ops.DeferredLayerNorm(a, b, ln)
        assert (again - want[2]).abs().max() < 2e-5 * float(want[2].abs().max())
        assert (d2.materialize() - y).abs().max() < 2e-5 and d2.done


def test_linear_split_bf16_grouped_adds_position_rows_on_load():
    """egtr_linear_split_bf16_grouped_pos_f32: a problem with ``pos`` multiplies (x + pos[row % pos_rows]) -- bit-identical
    to handing it the materialised sum; the other problem of the launch is untouched."""
    from egtr_amd import ops
    rng = W.rng_inputs(654)
    B, S, K = 2, 6300, 256
    x = torch.from_numpy(rng.standard_normal((B, S, K))).float().to(DEV)
    pos = torch.from_numpy(rng.standard_normal((S, K))).float().to(DEV)
    ws = [torch.from_numpy(rng.standard_normal((n, K)) / 16).float().to(DEV) for n in (256, 384)]
    bs = [torch.from_numpy(rng.standard_normal(n) * 0.1).float().to(DEV) for n in (256, 384)]
    wts = [ops.gemm_split_weights(w) for w in ws]
    want = ops.linear_split_bf16_grouped([dict(x=x, wt=wts[0], N=256, b=bs[0]), dict(x=x + pos, wt=wts[1], N=384, b=bs[1])])
    got = ops.linear_split_bf16_grouped([dict(x=x, wt=wts[0], N=256, b=bs[0]), dict(x=x, wt=wts[1], N=384, b=bs[1], pos=pos)])
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


@pytest.mark.parametrize("M,K,N,relu,bias", [(4800, 256, 256, False, True), (4800, 256, 1024, True, True), (4800, 1024, 256, False, True),
                                            (37, 256, 384, False, True), (1, 16, 32, True, False), (333, 48, 96, False, True)])
def test_linear_bf16_small_rows(M, K, N, relu, bias):
    """egtr_linear_bf16 (object-query-sized linears of a bf16 model) against the fp64 product of the same bf16 operands: fp32
    accumulation and ONE rounding of the result (half a bf16 ulp); ragged row tiles, N = 32 (one column tile) and K tails."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, generator=g).bfloat16() if bias else None
    want = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    if relu:
        want = want.relu()
    with torch.no_grad():
        got = ops.linear(x.to(DEV).view(1, M, K), w.to(DEV), b.to(DEV) if bias else None, relu=relu)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (1, M, N)
    err = (got[0].cpu().double() - want).abs() / want.abs().clamp_min(1.0)
    assert float(err.max()) < 2.0 ** -8 + 1e-4


@pytest.mark.parametrize("M,F,prow", [(256, 1024, 256), (700, 1024, 350), (33, 64, 0), (22223, 1024, 22223), (5, 32, 5)])
def test_ffn_layernorm_bf16_fused(M, F, prow):
    """egtr_ffn_layernorm_bf16 (fc1 + ReLU + fc2 + residual + LayerNorm (+ position rows) of a bf16 model, one launch) against
    (a) the torch bf16 composition of the same modules -- same rounding points, different accumulation order: within 2 bf16
    ulps of an O(1) LayerNorm output, and (b) the fp64 block: a looser bound that a layout mistake (O(1)) cannot meet."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(M + F)
    fc1, fc2, ln = torch.nn.Linear(256, F), torch.nn.Linear(F, 256), torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g)); ln.bias.copy_(0.2 * torch.randn(256, generator=g))
        fc1.bias.copy_(0.3 * torch.randn(F, generator=g)); fc2.bias.copy_(0.3 * torch.randn(256, generator=g))
    fc1, fc2, ln = fc1.to(DEV).bfloat16(), fc2.to(DEV).bfloat16(), ln.to(DEV).bfloat16()
    x = torch.randn(1, M, 256, generator=g).bfloat16().to(DEV)
    pos = torch.randn(prow, 256, generator=g).bfloat16().to(DEV) if prow else None
    with torch.no_grad():
        assert ops.ffn_bf16_supported(x, fc1, fc2, ln)
        out = ops.ffn_layernorm_bf16(x, fc1, fc2, ln, pos)
        y, yp = out if prow else (out, None)
        want = ln(x + fc2(torch.relu(fc1(x))))
        x64 = x.double()
        h64 = torch.relu(x64 @ fc1.weight.double().t() + fc1.bias.double())
        w64 = torch.nn.functional.layer_norm(x64 + h64 @ fc2.weight.double().t() + fc2.bias.double(), (256,),
                                             ln.weight.double(), ln.bias.double(), ln.eps)
    assert y.dtype == torch.bfloat16 and y.shape == x.shape
    scale = want.float().abs().clamp_min(1.0)
    assert float(((y.float() - want.float()).abs() / scale).max()) < 2.0 ** -6
    assert float(((y.double() - w64).abs() / w64.abs().clamp_min(1.0)).max()) < 3e-2
    if prow:
        assert torch.equal(yp, y + pos.repeat(M // prow, 1).view(1, M, 256))


@pytest.mark.parametrize("B,N", [(2, 300), (16, 300), (1, 37), (3, 200)])
def test_self_attention_bf16_entry_equals_the_fp32_kernel_on_widened_operands(B, N):
    """egtr_self_attn_forward_bf16 (round 6: the bf16 model's decoder without cast launches around the fp32 kernel; reference
    math dd:1170-1253): the SAME arithmetic in the same order on the exactly widened operands, so the output is the fp32
    kernel's output rounded to bf16 once (bit-identical), and the retained maps are bit copies of q / k in [B, M, N, D]."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N)
    q, k, v = ((torch.randn(B, N, 256, generator=g) * s).bfloat16().to(DEV) for s in (0.4, 1.0, 1.0))
    out, qh, kh = ops.decoder_self_attention(q, k, v, 8, want_maps=True)
    assert out.dtype == qh.dtype == kh.dtype == torch.bfloat16
    ref, rq, rk = ops.DecoderSelfAttentionFunction.apply(q.float(), k.float(), v.float(), 8, True)
    assert torch.equal(out, ref.bfloat16())
    assert torch.equal(qh, rq.bfloat16()) and torch.equal(kh, rk.bfloat16())
    assert torch.equal(qh, q.view(B, N, 8, 32).transpose(1, 2))
    out2, none_q, none_k = ops.decoder_self_attention(q, k, v, 8, want_maps=False)
    assert torch.equal(out2, out) and none_q is None and none_k is None


# ---- the bottleneck-tail kernel (csrc/conv_tail_x6.hip, egtr_conv1x1_tail_x6_f32) ------------------------------------------
def _tail_ref(a, sh, w, b, sc, relu_in, relu_out):
    x = a.double()
    if sh is not None:
        x = x + sh.double()
    if relu_in:
        x = torch.relu(x)
    y = x @ w.double().t()
    if b is not None:
        y = y + b.double()
    if sc is not None:
        y = y + sc.double()
    return torch.relu(y) if relu_out else y


@pytest.mark.parametrize("K,N", [(64, 256), (128, 512), (256, 1024), (512, 2048), (64, 128), (256, 384), (64, 64), (256, 192)])
@pytest.mark.parametrize("M", [1, 37, 608, 2399])
def test_conv1x1_tail_matches_fp64_product(K, N, M):
    """relu(relu(a + shift2) W3^T + shift3 + shortcut) in one launch against the fp64 product of the same fp32 operands: the
    six-term split-bf16 arithmetic has the error of an fp32 GEMM (1e-5 of the row's scale here); ragged row counts (the last
    panel is partial), every tile the dispatcher can pick, N that is a multiple of 128 but not of 256, and of 64 but not of 128 (two waves)."""
    from egtr_amd import ops
    torch.manual_seed(K + N + M)
    a = torch.randn(M, K, device=DEV)
    sh = torch.randn(K, device=DEV) * 0.3
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV) * 0.3
    sc = torch.randn(M, N, device=DEV)
    wxs = ops.xs_split(w, weights=True)
    ref = _tail_ref(a, sh, w, b, sc, True, True)
    tiles = [(0, 0)] + ([(32, 128)] if N % 128 == 0 else []) + ([(32, 256)] if N % 256 == 0 else []) \
        + ([(64, 128)] if K <= 256 and N % 128 == 0 else []) + ([(64, 256)] if K <= 256 and N % 256 == 0 else [])
    outs = []
    for tile in tiles:
        y = ops.conv1x1_tail(a, sh, wxs, b, sc, N, tile=tile)
        assert y.shape == (M, N) and y.dtype == torch.float32
        assert float((y.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), tile
        outs.append(y)
    # the tile only changes who computes an element, not how: same k order, same six terms
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert torch.equal(outs[0], ops.conv1x1_tail(a, sh, wxs, b, sc, N))   # run-to-run bit-identical (no atomics)


def test_conv1x1_tail_optional_operands_and_strides():
    """Every optional operand absent / present, both ReLUs off, and row strides larger than the row (a column block of a wider
    buffer as input, shortcut and -- through the view the caller makes -- output)."""
    from egtr_amd import ops
    torch.manual_seed(5)
    M, K, N = 333, 128, 256
    wide_a = torch.randn(M, K + 64, device=DEV)
    wide_s = torch.randn(M, N + 128, device=DEV)
    a, sc = wide_a[:, 32:32 + K], wide_s[:, 64:64 + N]
    assert a.data_ptr() % 16 == 0 and sc.data_ptr() % 16 == 0
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    wxs = ops.xs_split(w, weights=True)
    sh, b = torch.randn(K, device=DEV), torch.randn(N, device=DEV)
    for use_sh, use_b, use_sc, r_in, r_out in [(0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (0, 1, 0, 0, 1), (0, 0, 1, 1, 1), (1, 1, 1, 0, 0),
                                               (1, 1, 1, 1, 1)]:
        y = ops.conv1x1_tail(a, sh if use_sh else None, wxs, b if use_b else None, sc if use_sc else None, N,
                             relu_in=bool(r_in), relu_out=bool(r_out))
        ref = _tail_ref(a, sh if use_sh else None, w, b if use_b else None, sc if use_sc else None, r_in, r_out)
        assert float((y.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), (use_sh, use_b, use_sc, r_in, r_out)


def test_conv1x1_tail_non_finite_rows_stay_in_their_rows_and_bad_shapes_are_refused():
    from egtr_amd import ops
    from egtr_amd._lib import EgtrHipError
    torch.manual_seed(6)
    M, K, N = 100, 64, 256
    a = torch.randn(M, K, device=DEV)
    a[7, 3] = float("nan")
    a[50, 0] = float("inf")
    w = torch.randn(N, K, device=DEV) / 8
    wxs = ops.xs_split(w, weights=True)
    y = ops.conv1x1_tail(a, None, wxs, None, None, N, relu_in=False, relu_out=False)
    bad = ~torch.isfinite(y).all(dim=1)
    assert bad[7] and bad[50] and int(bad.sum()) == 2          # nothing leaks into the other rows of the panel
    # ReLU as torch computes it: relu(NaN) = NaN
    y2 = ops.conv1x1_tail(a, None, wxs, None, None, N, relu_in=True, relu_out=True)
    assert torch.isnan(y2[7]).all() and torch.isfinite(y2[6]).all()
    assert not ops.conv1x1_tail_supported(torch.randn(M, 96, device=DEV), 256)     # K not a bottleneck width
    assert not ops.conv1x1_tail_supported(a, 96)                                    # N not a multiple of 64
    with pytest.raises(EgtrHipError):
        ops.conv1x1_tail(torch.randn(M, 96, device=DEV), None, wxs, None, None, 256)
    with pytest.raises(EgtrHipError):
        ops.conv1x1_tail(a, None, wxs, None, None, N, tile=(64, 512))


@pytest.mark.parametrize("K,N", [(64, 256), (128, 512), (256, 1024), (512, 2048)])
@pytest.mark.parametrize("M", [1, 70, 2399])
def test_conv1x1_tail_bf16_matches_pass_gemm_pass(K, N, M):
    """The bf16 twin (csrc/conv_tail_bf16.hip) against the composition it replaces, with the same rounding points: shift + ReLU
    pass (bf16 out), bf16 GEMM with fp32 accumulation (bf16 out), shift + shortcut + ReLU pass.  Only the accumulation order of
    the product differs, so outputs agree except where a product lands within rounding distance of a bf16 tie: at most one
    bf16 ulp there, on a small fraction of the elements.  Also against the fp64 product with the same rounding points."""
    from egtr_amd import ops
    torch.manual_seed(K + M)
    a = torch.randn(M, K, device=DEV).bfloat16()
    sh = torch.randn(K, device=DEV) * 0.3
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=DEV) * 0.3
    sc = torch.randn(M, N, device=DEV).bfloat16()
    y = ops.conv1x1_tail_bf16(a, sh, ops.conv_tail_pack_bf16(w), b, sc, N)
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    a2 = a.clone()
    ops.bias_act_rows_(a2, sh)
    z = torch.mm(a2, w.t())
    ops.bias_act_rows_(z, b, sc)
    diff = (y.float() - z.float()).abs()
    ulp = torch.maximum(z.float().abs(), torch.tensor(1.0, device=DEV)) * 2.0 ** -7
    assert bool((diff <= ulp).all())
    assert float((diff > 0).float().mean()) < 0.02
    # fp64 product, same rounding points
    a64 = torch.relu(a.double() + sh.double()).bfloat16().double()
    z64 = (a64 @ w.double().t()).bfloat16().double()
    ref = torch.relu(z64 + b.double() + sc.double())
    assert float((y.double() - ref).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    assert torch.equal(y, ops.conv1x1_tail_bf16(a, sh, ops.conv_tail_pack_bf16(w), b, sc, N))


def test_conv1x1_tail_bf16_optional_operands_strides_and_refusals():
    from egtr_amd import ops
    from egtr_amd._lib import EgtrHipError
    torch.manual_seed(8)
    M, K, N = 333, 128, 256
    wide_a = torch.randn(M, K + 64, device=DEV).bfloat16()
    wide_s = torch.randn(M, N + 128, device=DEV).bfloat16()
    a, sc = wide_a[:, 32:32 + K], wide_s[:, 64:64 + N]
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).bfloat16()
    wp = ops.conv_tail_pack_bf16(w)
    sh, b = torch.randn(K, device=DEV), torch.randn(N, device=DEV)
    for use_sh, use_b, use_sc, r_in, r_out in [(0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (0, 1, 1, 0, 1), (1, 1, 1, 1, 1)]:
        y = ops.conv1x1_tail_bf16(a, sh if use_sh else None, wp, b if use_b else None, sc if use_sc else None, N,
                                  relu_in=bool(r_in), relu_out=bool(r_out))
        x = a.double() + (sh.double() if use_sh else 0.0)
        x = (torch.relu(x) if r_in else x).bfloat16().double()
        z = (x @ w.double().t()).bfloat16().double() + (b.double() if use_b else 0.0) + (sc.double() if use_sc else 0.0)
        ref = torch.relu(z) if r_out else z
        assert float((y.double() - ref).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref.abs().max())), (use_sh, use_b, use_sc)
    an = a.clone()
    an[5, 1] = float("nan")
    yn = ops.conv1x1_tail_bf16(an, None, wp, None, None, N, relu_in=True, relu_out=True)
    bad = ~torch.isfinite(yn.float()).all(dim=1)
    assert bad[5] and int(bad.sum()) == 1
    assert not ops.conv1x1_tail_bf16_supported(a, 128)
    with pytest.raises(EgtrHipError):
        ops.conv1x1_tail_bf16(a, None, wp, None, None, 128)


# ---- the split-bf16 3x3 convolution (csrc/conv3x3_x6.hip, egtr_conv3x3_x6_f32) ---------------------------------------------
@pytest.mark.parametrize("C,stride", [(64, 1), (128, 1), (256, 1), (512, 1), (128, 2), (256, 2), (512, 2)])
@pytest.mark.parametrize("B,H,W", [(1, 38, 63), (2, 7, 9), (1, 1, 1), (1, 4, 8), (3, 17, 33), (1, 8, 16)])
def test_conv3x3_x6_matches_fp64_convolution(C, stride, B, H, W):
    """3x3 / stride 1 or 2 / padding 1 on channels-last fp32 tensors against torch's fp64 convolution of the same fp32 operands:
    the six-term split-bf16 arithmetic has the error of an fp32 convolution (K = 9 C products per output: 2e-5 of the output
    scale here; MIOpen's fp32 kernels measure 2-4e-6, this kernel 4-9e-6).  Sizes that are not multiples of the pixel tiles, odd
    sizes at stride 2, images smaller than a tile, batches; every tile / phase variant; the padding is zeros."""
    import torch.nn.functional as F
    from egtr_amd import ops
    torch.manual_seed(C + H + W)
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device=DEV) / (9 * C) ** 0.5
    assert ops.conv3x3_supported(x, C, stride)
    ref = F.conv2d(x.double(), w.double(), None, stride=stride, padding=1)
    outs = []
    for variant in (0, 1, 2, 3, 4):
        if variant and not (variant in ({64: (1, 2, 3, 4), 128: (1, 2, 3, 4), 256: (1, 3), 512: (1,)} if stride == 1 else {128: (1,), 256: (1,), 512: (1,)}).get(C, ())):
            continue
        y = ops.conv3x3(x, ops.conv3x3_weights(w, stride, variant), C, stride, variant)
        assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
        assert float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), variant
        outs.append((variant, y))
    # a tile decides who computes an element, not how: the variants that walk K in one piece agree bit for bit (the ones that
    # split K over the waves or walk channel phases sum in another order)
    whole_k = {64: (0, 1, 2, 3), 128: (1, 2, 3, 4)}.get(C, ()) if stride == 1 else ()
    same = [o for v, o in outs if v in whole_k]
    assert all(torch.equal(same[0], o) for o in same[1:])
    assert torch.equal(outs[0][1], ops.conv3x3(x, ops.conv3x3_weights(w, stride), C, stride))   # run-to-run bit-identical


def test_conv3x3_x6_non_finite_inputs_and_refusals():
    import torch.nn.functional as F
    from egtr_amd import ops
    from egtr_amd._lib import EgtrHipError
    torch.manual_seed(9)
    x = torch.randn(1, 64, 12, 20, device=DEV).contiguous(memory_format=torch.channels_last)
    x[0, 5, 6, 7] = float("nan")
    w = torch.randn(64, 64, 3, 3, device=DEV) / 24
    y = ops.conv3x3(x, ops.conv3x3_weights(w), 64)
    y2 = ops.conv3x3(x, ops.conv3x3_weights(torch.randn(128, 128, 3, 3, device=DEV)[:64, :64].contiguous(), 1), 64)
    assert y2.shape == y.shape
    bad = ~torch.isfinite(y).all(dim=1)[0]                      # [H, W]: pixels with a non-finite channel
    want = torch.zeros(12, 20, dtype=torch.bool, device=DEV)
    want[5:8, 6:9] = True                                       # exactly the 3 x 3 neighbourhood that reads the NaN
    assert torch.equal(bad, want)
    ref = F.conv2d(torch.nan_to_num(x), w, None, padding=1)
    assert float((y - ref)[0, :, ~want].abs().max()) < 1e-4
    assert not ops.conv3x3_supported(x.contiguous(), 64)                                   # NCHW memory
    assert not ops.conv3x3_supported(torch.randn(1, 96, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last), 96)
    with pytest.raises(EgtrHipError):
        ops.conv3x3(torch.randn(1, 96, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last), ops.conv3x3_weights(w), 96)


# ---- the fused stem (csrc/stem_x6.hip, egtr_stem_conv7x7_pool_x6_f32) -------------------------------------------------------
@pytest.mark.parametrize("B,H,W", [(1, 224, 320), (2, 61, 83), (1, 7, 9), (1, 1, 1), (1, 64, 128), (3, 33, 37)])
def test_stem_fused_matches_fp64_conv_relu_pool(B, H, W):
    """7x7/2 convolution + shift + ReLU + 3x3/2 max-pool in one launch, channels-last out, against the fp64 composition of the
    same fp32 operands (six-term split-bf16 products: 1e-5 of the output scale); odd sizes, images smaller than a tile (the
    zero padding of both the convolution and the pool is exercised everywhere at the borders), batches."""
    import torch.nn.functional as F
    from egtr_amd import ops
    torch.manual_seed(H + W)
    x = torch.randn(B, 3, H, W, device=DEV)
    w = torch.randn(64, 3, 7, 7, device=DEV) / 147 ** 0.5
    b = torch.randn(64, device=DEV) * 0.3
    assert ops.stem_fused_supported(x, w)
    y = ops.stem_fused(x, ops.stem_weights(w), b)
    ref = F.max_pool2d(torch.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3) + b.double().view(1, -1, 1, 1)), 3, 2, 1)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(y, ops.stem_fused(x, ops.stem_weights(w), b))


def test_stem_fused_nan_pixel_reaches_exactly_its_pool_windows():
    import torch.nn.functional as F
    from egtr_amd import ops
    torch.manual_seed(11)
    x = torch.randn(1, 3, 40, 56, device=DEV)
    x[0, 1, 20, 30] = float("nan")
    w = torch.randn(64, 3, 7, 7, device=DEV) / 12
    b = torch.zeros(64, device=DEV)
    y = ops.stem_fused(x, ops.stem_weights(w), b)
    # (fp64 reference: a direct convolution -- MIOpen's fp32 Winograd kernel spreads a NaN over its whole transform tile)
    ref = F.max_pool2d(torch.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3)), 3, 2, 1)
    assert torch.equal(torch.isnan(y), torch.isnan(ref))          # exactly the windows that contain the pixel, all 64 channels
    assert int(torch.isnan(ref).any(dim=1).sum()) in (4, 6, 9)   # a 7 x 7 window / stride 2, pooled 3 x 3 / stride 2
    ok = ~torch.isnan(ref)
    assert float((y[ok].double() - ref[ok]).abs().max()) < 1e-4


@pytest.mark.parametrize("C,N", [(256, 512), (512, 1024), (1024, 2048), (256, 128)])
@pytest.mark.parametrize("B,H,W,stride", [(1, 38, 63, 2), (2, 7, 9, 2), (1, 1, 1, 2), (1, 8, 16, 1), (2, 17, 33, 1)])
def test_conv1x1_strided_matches_fp64_convolution(C, N, B, H, W, stride):
    """The shortcut projection of a resolution-changing bottleneck (1x1 convolution, stride 2; stride 1 served as well) through
    the one-tap form of the own convolution kernel against torch's fp64 convolution: odd sizes (the last row / column is read
    or not), sizes below a tile, batches, every channel-phase count."""
    import torch.nn.functional as F
    from egtr_amd import ops
    torch.manual_seed(C + H + W)
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(N, C, 1, 1, device=DEV) / C ** 0.5
    assert ops.conv1x1_strided_supported(x, N, stride)
    y = ops.conv1x1_strided(x, ops.xs_split(w.reshape(N, C).contiguous(), weights=True), N, stride)
    ref = F.conv2d(x.double(), w.double(), None, stride=stride).permute(0, 2, 3, 1).reshape(-1, N)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(y, ops.conv1x1_strided(x, ops.xs_split(w.reshape(N, C).contiguous(), weights=True), N, stride))


@pytest.mark.parametrize("B,H,W", [(1, 224, 320), (2, 61, 83), (1, 7, 9), (1, 1, 1), (3, 33, 37)])
def test_stem_fused_bf16_matches_the_composition_it_replaces(B, H, W):
    """The bf16 stem kernel (csrc/stem_bf16.hip) against fp64 arithmetic with its rounding points -- bf16 inputs / weights, the
    convolution output rounded to bf16, shift + ReLU in fp32, max-pool, rounded once more -- to one bf16 ulp of the output scale
    (the accumulation order differs, so a product on a rounding tie may flip), and against torch's bf16 convolution + pool +
    the shift / ReLU pass it replaces."""
    import torch.nn.functional as F
    from egtr_amd import ops
    torch.manual_seed(H + W)
    x = torch.randn(B, 3, H, W, device=DEV).bfloat16()
    w = (torch.randn(64, 3, 7, 7, device=DEV) / 147 ** 0.5).bfloat16()
    b = torch.randn(64, device=DEV) * 0.3
    assert ops.stem_fused_bf16_supported(x, w)
    y = ops.stem_fused_bf16(x, ops.stem_weights_bf16(w), b)
    conv = F.conv2d(x.double(), w.double(), None, stride=2, padding=3).bfloat16().double()
    ref = F.max_pool2d(torch.relu(conv + b.double().view(1, -1, 1, 1)), 3, 2, 1).bfloat16()
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    scale = max(1.0, float(ref.float().abs().max()))
    assert float((y.double() - ref.double()).abs().max()) <= 2.0 ** -7 * scale
    assert float((y != ref).float().mean()) < 0.02
    # the route it replaces
    t = F.max_pool2d(F.conv2d(x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last),
                              None, stride=2, padding=3), 3, 2, 1).contiguous(memory_format=torch.channels_last)
    ops.bias_act_rows_(t.permute(0, 2, 3, 1).reshape(-1, 64), b)
    assert float((y.float() - t.float()).abs().max()) <= 2.0 ** -6 * scale
    assert torch.equal(y, ops.stem_fused_bf16(x, ops.stem_weights_bf16(w), b))
