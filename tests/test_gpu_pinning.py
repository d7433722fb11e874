"""GPU (-m gpu): what the headline number rests on, pinned DIRECTLY against the oracle ("Next round" item 1).

  * the three fused MSDA entries the timed forward calls (egtr_msda_forward_fused_vbias_f32 with and without the bias / _box_f32) against
    oracle.msda.msda_forward fed with the HOST-composed softmax / sampling locations / masked (+ biased) values
    (reference: model/deformable_detr.py:1048-1081), at the small pyramid and at S = Lq = 12 537 (sampled rows);
  * the relation / connectivity loss kernel against oracle.loss (egtr:754-923 restated, pinned to the reference fixtures);
  * the relation-head kernels against oracle.detr.relation_head -- the reference's evaluation ORDER
    (relation_source -> gate -> gated sum -> MLPs, egtr:322-418), not a separable restatement;
  * every split-bf16 ("fp32 via six bf16 cross terms") kernel on adversarial operands: rows spanning 2^-60 .. 2^60,
    cancelling sums, denormal inputs, activations ~1e4, +-inf / NaN rows.

Tolerances are written at each assert; "2.5x" means: no further from float64 than 2.5 times the exact-fp32 evaluation of
the same product (vendor fp32 GEMM / exact-f32 MFMA kernel) on the same operands.
"""
import numpy as np
import pytest
import torch

import weights as W
from oracle import detr as O
from oracle import loss as OL
from oracle import msda as OM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SMALL = [(19, 32), (10, 16), (5, 8), (3, 4)]
FULL = [(75, 125), (38, 63), (19, 32), (10, 16)]   # the 600 x 1000 pyramid, S = 12 537


def _kernels():
    from egtr_amd.load_custom import load_hip_kernels
    return load_hip_kernels()


def _fused_case(seed, B, shapes, Lq, ref_dim, off_scale):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    Lq = S if Lq is None else Lq
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, 8, 32, generator=g)
    bias = torch.randn(256, generator=g)
    off = torch.randn(B, Lq, 8, 4, 4, 2, generator=g) * off_scale
    logits = torch.randn(B, Lq, 8, 16, generator=g) * 2
    if ref_dim == 2:
        ref = torch.rand(B, Lq, 4, 2, generator=g) * 1.1 - 0.05        # a few reference points outside [0, 1]
    else:
        ref = torch.cat([torch.rand(B, Lq, 4, 2, generator=g), torch.rand(B, Lq, 4, 2, generator=g) * 0.5 + 0.05], -1)
    keep = torch.rand(B, S, generator=g) > 0.25                        # 25 % padded tokens
    return dict(shp=shp, lsi=lsi, value=value, bias=bias, off=off, logits=logits, ref=ref, keep=keep, S=S, Lq=Lq)


def _host_composition(c, rows, use_bias, use_mask):
    """dd:1048-1081 on the host for the sampled query rows: (masked, biased) values, softmax, sampling locations."""
    value = c["value"]
    if use_bias:
        value = value + c["bias"].view(1, 1, 8, 32)
    if use_mask:
        value = torch.where(c["keep"][..., None, None], value, torch.zeros(()))
    off, logits, ref = c["off"][:, rows], c["logits"][:, rows], c["ref"][:, rows]
    attn = torch.softmax(logits, -1).view(*logits.shape[:3], 4, 4)
    if ref.shape[-1] == 2:
        norm = torch.stack([c["shp"][:, 1], c["shp"][:, 0]], -1).float()          # (W, H)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / 4 * ref[:, :, None, :, None, 2:] * 0.5
    return value, loc.contiguous(), attn.contiguous()


@pytest.mark.parametrize("name,B,shapes,Lq,ref_dim", [
    ("encoder-small", 2, SMALL, None, 2),       # Lq = S: the wave-per-query kernel
    ("decoder-split", 2, SMALL, 300, 2),        # B * Lq <= 1024: one workgroup per query, samples split over its waves
    ("decoder-large", 1, SMALL, 1400, 2),
    ("box-refs", 2, SMALL, 300, 4),             # 4-d reference boxes (iterative refinement)
    ("box-refs-enc", 1, SMALL, None, 4),
    ("encoder-600x1000", 1, FULL, None, 2),     # the bench's launch shape: S = Lq = 12 537
    ("encoder-600x1000-b2", 2, FULL, None, 2),
])
def test_fused_msda_entries_vs_oracle(name, B, shapes, Lq, ref_dim):
    """2e-5 on O(1) outputs; ~5 % of the samples fall outside the maps (offset scale), 25 % of the tokens are padding;
    every combination of {value bias in the kernel, padding mask in the kernel}; the returned softmax weights too."""
    k = _kernels()
    c = _fused_case(40 + len(name) + B, B, shapes, Lq, ref_dim, off_scale=2.0 if shapes is FULL else 1.0)
    S, Lq = c["S"], c["Lq"]
    rows = torch.arange(Lq) if Lq <= 1500 else torch.from_numpy(
        np.sort(W.rng_inputs(5).choice(Lq, 500, replace=False))).long()
    rows = torch.unique(torch.cat([rows, torch.tensor([0, Lq - 1])]))
    d = {n: c[n].to(DEV) for n in ("shp", "lsi", "value", "bias", "off", "logits", "ref", "keep")}
    oob = None
    for use_bias in (False, True):
        for use_mask in (False, True):
            out, wts = k.ms_deform_attn_forward_fused(
                d["value"], d["shp"], d["lsi"], d["off"], d["logits"], d["ref"], True, d["keep"] if use_mask else None,
                value_bias=d["bias"] if use_bias else None)
            value, loc, attn = _host_composition(c, rows, use_bias, use_mask)
            want = OM.msda_forward(value, c["shp"], c["lsi"], loc, attn)
            err = (out.cpu()[:, rows] - want).abs().max()
            assert err < 2e-5, (name, use_bias, use_mask, float(err))
            assert (wts.cpu()[:, rows] - attn).abs().max() < 1e-6
            oob = float(((loc < 0) | (loc > 1)).any(-1).float().mean())
    assert 0.01 < oob < 0.5, oob        # the case really exercises out-of-range samples


def test_fused_msda_bitpacked_mask_and_strided_operands_vs_oracle():
    """The forward's own calling convention: offsets / logits as column blocks of ONE [B, Lq, 384] projection output, the
    padding mask bit-packed by the level-geometry kernel -- against the oracle, not against another fused call."""
    k = _kernels()
    c = _fused_case(77, 2, SMALL, None, 2, 1.0)
    B, S = 2, c["S"]
    both = torch.cat([c["off"].reshape(B, S, 256), c["logits"].reshape(B, S, 128)], -1).to(DEV)
    off = both[..., :256].view(B, S, 8, 4, 4, 2)
    logits = both[..., 256:].view(B, S, 8, 16)
    words = torch.zeros(B, (S + 31) // 32, dtype=torch.int64)
    idx = torch.arange(S)
    for bi in range(B):
        words[bi].index_add_(0, idx // 32, c["keep"][bi].long() << (idx % 32))
    km = c["keep"].to(DEV)
    km._egtr_bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).to(DEV)
    out, _ = k.ms_deform_attn_forward_fused(c["value"].to(DEV), c["shp"].to(DEV), c["lsi"].to(DEV), off, logits,
                                            c["ref"].to(DEV), False, km, value_bias=c["bias"].to(DEV))
    value, loc, attn = _host_composition(c, torch.arange(S), True, True)
    want = OM.msda_forward(value, c["shp"], c["lsi"], loc, attn)
    assert (out.cpu() - want).abs().max() < 2e-5


# ------------------------------------------------------------------------------------------------ relation loss
@pytest.mark.parametrize("B,N,R,Ts,nrel", [
    (3, 40, 7, (5, 12, 1), 3),        # small: k1 limited by the number of false candidates
    (2, 200, 50, (30, 9), 3),         # VG-sized
    (2, 64, 9, (6, 0), 2),            # an image without targets
    (1, 200, 50, (17,), 0),           # no relation at all: mean of an empty tensor (NaN), zero gradients
])
def test_relation_loss_kernel_vs_oracle(B, N, R, Ts, nrel):
    """egtr_relation_loss_f32 (value + dense gradient) against oracle.loss.relation_losses -- the restatement of
    egtr:754-923 that tests/test_oracle_golden.py pins to the reference's loss dicts -- in float64 under autograd.
    Value 1e-5 relative, gradient 1e-7 absolute."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(17 + N)
    nm_cost = OL.nonmatching_cost(2.0, 5.0, 2.0, 1e-14)
    pred_rel = torch.randn(B, N, N, R, generator=g) * 2
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
        indices.append((torch.randperm(N, generator=g)[:T].sort()[0], torch.randperm(T, generator=g)))
        costs.append(torch.randn(T, generator=g) * 3)
    pr64 = pred_rel.double().requires_grad_(True)
    pc64 = pred_conn.double().requires_grad_(True)
    w_rel, w_conn = OL.relation_losses(pr64, pc64, [{"rel": t["rel"].double()} for t in targets], indices,
                                       [c.double() for c in costs], nm_cost, 80, 80, True)
    nan_case = bool(torch.isnan(w_rel))
    ((w_conn * 0.5) if nan_case else (w_rel * 1.5 + w_conn * 0.5)).backward()
    prd = pred_rel.to(DEV).requires_grad_(True)
    pcd = pred_conn.to(DEV).requires_grad_(True)
    l_rel, l_conn = ops.relation_losses(prd, pcd, [{"rel": t["rel"].to(DEV)} for t in targets],
                                        [(a.to(DEV), b_.to(DEV)) for a, b_ in indices], [c.to(DEV) for c in costs],
                                        float(nm_cost), 80, 80)
    assert abs(float(l_conn) - float(w_conn)) < 1e-5 * max(1.0, abs(float(w_conn)))
    if nan_case:
        assert bool(torch.isnan(l_rel))
        (l_conn * 0.5).backward()
        assert prd.grad is None or float(prd.grad.abs().max()) == 0.0
    else:
        assert abs(float(l_rel) - float(w_rel)) < 1e-5 * max(1.0, abs(float(w_rel)))
        (l_rel * 1.5 + l_conn * 0.5).backward()
        assert (prd.grad.cpu().double() - pr64.grad).abs().max() < 1e-7
    assert (pcd.grad.cpu().double() - pc64.grad).abs().max() < 1e-7


# ------------------------------------------------------------------------------------------------ relation head
def _reference_form_head(seed, B, N, Ld, R, C):
    """Random parameters / inputs of the relation head in the REFERENCE's own form (state-dict names of egtr.py)."""
    rng = W.rng_inputs(seed)
    r = lambda *s, sc=1.0: torch.from_numpy(rng.standard_normal(s) * sc).double()  # noqa: E731
    d = 256
    sd = {"triplet_dist": r(C + 1, C + 1, R)}
    for l in range(Ld):
        for n in ("proj_q", "proj_k"):
            sd[f"{n}.{l}.weight"], sd[f"{n}.{l}.bias"] = r(d, d, sc=1 / 16), r(d, sc=0.1)
    for n in ("final_sub_proj", "final_obj_proj"):
        sd[f"{n}.weight"], sd[f"{n}.bias"] = r(d, d, sc=1 / 16), r(d, sc=0.1)
    sd["rel_predictor_gate.weight"], sd["rel_predictor_gate.bias"] = r(1, 2 * d, sc=1 / 16), r(1, sc=0.1)
    for n, out in (("rel_predictor", R), ("connectivity_layer", 1)):
        sd[f"{n}.layers.0.weight"], sd[f"{n}.layers.0.bias"] = r(d, 2 * d, sc=1 / 22), r(d, sc=0.1)
        sd[f"{n}.layers.1.weight"], sd[f"{n}.layers.1.bias"] = r(d, d, sc=1 / 16), r(d, sc=0.1)
        sd[f"{n}.layers.2.weight"], sd[f"{n}.layers.2.bias"] = r(out, d, sc=1 / 16), r(out, sc=0.1)
    queries = [r(B, 8, N, 32, sc=0.2) for _ in range(Ld)]      # scaled q (dd:1166) and k maps [B, M, N, D]
    keys = [r(B, 8, N, 32) for _ in range(Ld)]
    hidden = r(B, N, d)
    logits = r(B, N, C)
    return sd, queries, keys, hidden, logits


def _kernel_tables(sd, queries, keys, hidden):
    """The kernel's per-query tables from the reference-form parameters, in float64 (exact separable algebra:
    w_g.[q;k] = a_i + c_j and W1.[q;k] = Uq_i + Uk_j), rounded to fp32 once."""
    B, N, d = hidden.shape
    unscale = 32 ** 0.5
    lin = lambda n, x: x @ sd[f"{n}.weight"].t() + sd[f"{n}.bias"]  # noqa: E731
    pq = [lin(f"proj_q.{l}", q.transpose(1, 2).reshape(B, N, d) * unscale) for l, q in enumerate(queries)]
    pk = [lin(f"proj_k.{l}", k.transpose(1, 2).reshape(B, N, d)) for l, k in enumerate(keys)]
    Q = torch.stack(pq + [lin("final_sub_proj", hidden)], -2)
    K = torch.stack(pk + [lin("final_obj_proj", hidden)], -2)
    w1 = torch.cat([sd["rel_predictor.layers.0.weight"], sd["connectivity_layer.layers.0.weight"]], 0)
    wg = sd["rel_predictor_gate.weight"]
    t = dict(gate_q=Q @ wg[0, :d], gate_k=K @ wg[0, d:] + sd["rel_predictor_gate.bias"], uq=Q @ w1[:, :d].t(),
             uk=K @ w1[:, d:].t(),
             b1=torch.cat([sd["rel_predictor.layers.0.bias"], sd["connectivity_layer.layers.0.bias"]]),
             w2r=sd["rel_predictor.layers.1.weight"], b2r=sd["rel_predictor.layers.1.bias"],
             w3r=sd["rel_predictor.layers.2.weight"], b3r=sd["rel_predictor.layers.2.bias"],
             w2c=sd["connectivity_layer.layers.1.weight"], b2c=sd["connectivity_layer.layers.1.bias"],
             w3c=sd["connectivity_layer.layers.2.weight"], b3c=sd["connectivity_layer.layers.2.bias"])
    return {k: v.float().contiguous().to(DEV) for k, v in t.items()}


@pytest.mark.parametrize("B,N,Ld,R", [(1, 200, 6, 50), (2, 24, 3, 7), (1, 100, 3, 30), (1, 33, 8, 64), (2, 37, 6, 32)])
def test_relation_head_kernels_vs_oracle_reference_order(B, N, Ld, R):
    """Both inference kernels (exact-f32 MFMA, split-bf16 x6) and the training forward against
    oracle.detr.relation_head in float64: the reference's order of evaluation with the materialised
    relation_source [B, N, N, Ld + 1, 512].  2e-4 on logits of magnitude O(1..10) (north-star: 1e-3); gate means 1e-5."""
    from egtr_amd import ops
    C = 11
    sd, queries, keys, hidden, logits = _reference_form_head(900 + N, B, N, Ld, R, C)
    cfg = dict(encoder_attention_heads=8, use_freq_bias=True)
    rrel, rconn, rgate = O.relation_head(sd, cfg, queries, keys, hidden, logits)
    t = _kernel_tables(sd, queries, keys, hidden)
    trip = sd["triplet_dist"].float().to(DEV)
    node = torch.argmax(logits, -1).to(DEV)
    rel32, conn32, gm32 = ops.RelationHeadFunction.apply(*t.values(), trip, node, True)
    w2xr, w3xr, w2xc = ops.rel_head_split_weights(t["w2r"], t["w3r"], t["w2c"])
    rel6, conn6, gm6 = ops.relation_head_split_bf16(
        t["gate_q"], t["gate_k"], t["uq"], t["uk"], t["b1"], w2xr, t["b2r"], w3xr, t["b3r"], w2xc, t["b2c"], t["w3c"],
        t["b3c"], R, trip, node, True)
    want_gm = rgate.reshape(-1, Ld + 1).mean(0)
    for rel, conn, gm in ((rel32, conn32, gm32), (rel6, conn6, gm6)):
        assert (rel.cpu().double() - rrel).abs().max() < 2e-4
        assert (conn.cpu().double() - rconn).abs().max() < 2e-4
        assert (gm.cpu().double() - want_gm).abs().max() < 1e-5


# ------------------------------------------------------------------------------ split-bf16 arithmetic, adversarial operands
def _errs(got, ref):
    e = (got.double().cpu() - ref).abs()
    return float(e.max()), float(e.norm())


def _adversarial_rows(kind, M, K, N, rng):
    """(x [M, K], w [N, K]) fp32 CPU tensors for one adversarial family."""
    x = torch.from_numpy(rng.standard_normal((M, K))).float()
    w = torch.from_numpy(rng.standard_normal((N, K)) / np.sqrt(K)).float()
    if kind == "wide-range":        # every row of x and w at its own power-of-two scale: 2^-60 .. 2^60 / 2^-20 .. 2^20
        x = x * torch.pow(2.0, torch.from_numpy(rng.integers(-60, 61, (M, 1))).float())
        w = w * torch.pow(2.0, torch.from_numpy(rng.integers(-20, 21, (N, 1))).float())
    elif kind == "cancelling":      # sum |w x| >> |sum w x|: paired columns carry +-a with (almost) equal weights
        x[:, 1::2] = -x[:, 0::2] * (1 + 1e-6 * torch.from_numpy(rng.standard_normal((M, K // 2))).float())
        w[:, 1::2] = w[:, 0::2]
        x = x * 300
    elif kind == "large":           # activations ~1e4, as after an un-normalised FFN
        x = x * 1e4
    elif kind == "denormal":        # fp32 denormals (and values whose LAST piece is denormal)
        x = x * 1e-39
        x[::3] = x[::3] * 1e4
    return x, w


@pytest.mark.parametrize("kind", ["wide-range", "cancelling", "large", "denormal"])
def test_split_bf16_linears_on_adversarial_operands(kind):
    """egtr_linear_split_bf16_f32 and the training weight-gradient kernel against float64: row by row no further off than 2.5x the vendor fp32 GEMM (+ one ulp-sized floor relative to
    the row's sum |w| |x|), for operands far from N(0, 1).  Denormal inputs: the matrix pipe may flush denormal pieces;
    the bound is then ABSOLUTE, 2^-126 * K (nothing above the smallest normal is lost)."""
    from egtr_amd import ops
    M, K, N = 4224, 256, 256
    rng = W.rng_inputs(5000 + len(kind))
    x, w = _adversarial_rows(kind, M, K, N, rng)
    ref = x.double() @ w.double().t()
    mag = x.double().abs() @ w.double().abs().t()                 # sum |w| |x| per output
    xd, wd = x.to(DEV), w.to(DEV)
    y32 = torch.nn.functional.linear(xd, wd)
    y_old = ops.linear_split_bf16(xd, ops.gemm_split_weights(wd), None, N)
    e32 = (y32.double().cpu() - ref).abs()
    for name, y in (("gemm_split", y_old),):
        e = (y.double().cpu() - ref).abs()
        assert torch.isfinite(y).all(), name
        if kind == "denormal":
            assert (e <= 2.0 ** -126 * K + 2.0 ** -22 * mag).all(), (name, float(e.max()))
        else:
            # per output: fp32-level relative to sum |w||x| (2^-21: a K = 256 fp32 accumulation), and per row 2.5x vendor
            assert (e <= 2.0 ** -21 * mag + 1e-37).all(), (name, float((e / (mag + 1e-300)).max()))
            rows_e, rows_32 = e.max(1)[0], e32.max(1)[0]
            floor = 2.0 ** -23 * mag.max(1)[0]
            assert (rows_e <= 2.5 * rows_32 + floor).all(), (name, float((rows_e / (rows_32 + floor)).max()))
    # weight gradient g^T x over the rows (reduction over M): same operands, g = x-like
    g = x[:, :128].contiguous().to(DEV)
    gw = ops.linear_split_bf16_wgrad(g, xd)
    ref_w = g.double().cpu().t() @ x.double()
    mag_w = g.double().cpu().abs().t() @ x.double().abs()
    e_w = (gw.double().cpu() - ref_w).abs()
    e_w32 = ((g.t() @ xd).double().cpu() - ref_w).abs()
    if kind == "denormal":
        assert (e_w <= 2.0 ** -126 * M + 2.0 ** -20 * mag_w).all()
    else:
        assert (e_w <= 2.0 ** -18 * mag_w + 1e-37).all(), float((e_w / (mag_w + 1e-300)).max())
        assert e_w.max() <= 2.5 * e_w32.max() + 2.0 ** -22 * float(mag_w.max())


def test_split_bf16_linears_propagate_non_finite_rows_only():
    """A +-inf / NaN element makes exactly ITS output row non-finite (the vendor fp32 GEMM yields +-inf or NaN there; the
    split kernels yield NaN or +-inf -- hi carries the inf, the 0 x inf of the zeroed lower pieces is NaN: documented in
    DESIGN 4.12); every other row is bit-identical to the run without the poisoned elements."""
    from egtr_amd import ops
    M, K, N = 4224, 256, 384
    rng = W.rng_inputs(5100)
    x, w = _adversarial_rows("plain", M, K, N, rng)
    bad_rows = {5: float("inf"), 131: float("-inf"), 4000: float("nan"), 4223: float("inf")}
    xb = x.clone()
    for r, v in bad_rows.items():
        xb[r, (7 * r) % K] = v
    xd, xbd, wd = x.to(DEV), xb.to(DEV), w.to(DEV)
    wt = ops.gemm_split_weights(wd)
    good = torch.ones(M, dtype=torch.bool)
    good[list(bad_rows)] = False
    for name, fn in (("gemm_split", lambda t: ops.linear_split_bf16(t, wt, None, N)),):
        clean, dirty = fn(xd).cpu(), fn(xbd).cpu()
        assert torch.equal(clean[good], dirty[good]), name
        assert not torch.isfinite(dirty[~good]).any(), name      # the whole poisoned row is non-finite
    # weight gradient: the poisoned element x[r, k] spoils column k of g^T x for every n, nothing else
    g = torch.from_numpy(rng.standard_normal((M, 128))).float().to(DEV)
    gw_c, gw_d = ops.linear_split_bf16_wgrad(g, xd).cpu(), ops.linear_split_bf16_wgrad(g, xbd).cpu()
    cols = torch.ones(K, dtype=torch.bool)
    cols[[(7 * r) % K for r in bad_rows]] = False
    assert torch.equal(gw_c[:, cols], gw_d[:, cols])
    assert not torch.isfinite(gw_d[:, ~cols]).any()


@pytest.mark.parametrize("kind", ["large", "cancelling", "tiny", "non-finite"])
def test_relation_head_split_bf16_on_adversarial_operands(kind):
    """rel_head_fwd_x6 (layers 2 / 3 from split operands) vs the exact-f32 MFMA kernel, both against the float64
    restatement: hidden activations ~1e4 ("large"), paired +- columns in W2 ("cancelling"), activations ~1e-30 ("tiny"),
    and an inf / NaN in one subject's table ("non-finite": exactly the pairs of that subject become non-finite)."""
    import cpu_kernels as ck
    from egtr_amd import ops
    B, N, T, R = 1, 48, 4, 20
    rng = W.rng_inputs(5200 + len(kind))
    r = lambda *s, sc=1.0: torch.from_numpy(rng.standard_normal(s) * sc).float()  # noqa: E731
    d = dict(gate_q=r(B, N, T), gate_k=r(B, N, T), uq=r(B, N, T, 512, sc=0.5), uk=r(B, N, T, 512, sc=0.5),
             b1=r(512, sc=0.1), w2r=r(256, 256, sc=1 / 16), b2r=r(256, sc=0.1), w3r=r(R, 256, sc=1 / 16),
             b3r=r(R, sc=0.1), w2c=r(256, 256, sc=1 / 16), b2c=r(256, sc=0.1), w3c=r(1, 256, sc=1 / 16), b3c=r(1, sc=0.1))
    if kind == "large":
        d["uq"], d["uk"] = d["uq"] * 1e4, d["uk"] * 1e4
    elif kind == "tiny":
        for n in ("uq", "uk", "b1"):
            d[n] = d[n] * 1e-30
    elif kind == "cancelling":
        for n in ("w2r", "w2c"):
            d[n][:, 1::2] = -d[n][:, 0::2]
        for n in ("uq", "uk"):      # neighbouring hidden-1 channels (almost) equal: layer 2 sums +-pairs
            d[n][..., 1::2] = d[n][..., 0::2] * (1 + 1e-6)
        d["b1"][1::2] = d["b1"][0::2]
    poisoned = None
    if kind == "non-finite":
        poisoned = {3: float("inf"), 17: float("nan")}
        for i, v in poisoned.items():
            d["uq"][0, i, 1, 5 + i] = v
    dd = {k: v.to(DEV) for k, v in d.items()}
    w2xr, w3xr, w2xc = ops.rel_head_split_weights(dd["w2r"], dd["w3r"], dd["w2c"])
    fwd = ops.relation_head_split_bf16
    rel6, conn6, _ = fwd(
        dd["gate_q"], dd["gate_k"], dd["uq"], dd["uk"], dd["b1"], w2xr, dd["b2r"], w3xr, dd["b3r"], w2xc, dd["b2c"],
        dd["w3c"], dd["b3c"], R, None, None, False)
    rel32, conn32, _ = ops.RelationHeadFunction.apply(*dd.values(), None, None, False)
    if kind == "non-finite":
        ok = torch.ones(N, dtype=torch.bool)
        ok[list(poisoned)] = False
        clean = {k: v.clone() for k, v in d.items()}
        for i in poisoned:
            clean["uq"][0, i, 1, 5 + i] = 0.0
        cd = {k: v.to(DEV) for k, v in clean.items()}
        rel_c, conn_c, _ = fwd(
            cd["gate_q"], cd["gate_k"], cd["uq"], cd["uk"], cd["b1"], w2xr, cd["b2r"], w3xr, cd["b3r"], w2xc, cd["b2c"],
            cd["w3c"], cd["b3c"], R, None, None, False)
        assert torch.equal(rel6[0, ok], rel_c[0, ok]) and torch.equal(conn6[0, ok], conn_c[0, ok])
        # the poisoned hidden-1 channel (relation half; ReLU keeps NaN like torch.relu) reaches every relation output of
        # the subject's pairs, and nothing of the connectivity half
        assert not torch.isfinite(rel6[0, ~ok]).any()
        assert not torch.isfinite(rel32[0, ~ok]).any()
        assert torch.equal(conn6[0, ~ok], conn_c[0, ~ok])
        return
    d64 = {k: v.double() for k, v in d.items()}
    rrel, rconn, _ = ck.relation_head(*d64.values(), None, None, False)
    for got, got32, ref in ((rel6, rel32, rrel), (conn6, conn32, rconn)):
        e, e32 = (got.cpu().double() - ref).abs(), (got32.cpu().double() - ref).abs()
        scale = max(1.0, float(ref.abs().max()))
        assert torch.isfinite(got).all()
        assert e.max() <= 2.5 * e32.max() + 2e-6 * scale, (kind, float(e.max()), float(e32.max()), scale)
        assert e.norm() <= 2.5 * e32.norm() + 2e-6 * scale * e.numel() ** 0.5


# ------------------------------------------------------------------------------------------------ fused FFN (csrc/ffn_x6.hip)
def _ffn_modules(F, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    fc1, fc2, ln = torch.nn.Linear(256, F), torch.nn.Linear(F, 256), torch.nn.LayerNorm(256)
    with torch.no_grad():
        fc1.weight.copy_(torch.randn(F, 256, generator=g) / 16 * scale)
        fc1.bias.copy_(torch.randn(F, generator=g) * 0.3)
        fc2.weight.copy_(torch.randn(256, F, generator=g) / F ** 0.5)
        fc2.bias.copy_(torch.randn(256, generator=g) * 0.3)
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
        ln.bias.copy_(0.2 * torch.randn(256, generator=g))
    return fc1, fc2, ln


def _ffn_ref64(x, fc1, fc2, ln=None):
    x = x.double()
    h = torch.relu(x @ fc1.weight.double().t() + fc1.bias.double())
    y = h @ fc2.weight.double().t() + fc2.bias.double()
    if ln is None:
        return y
    return torch.nn.functional.layer_norm(x + y, (256,), ln.weight.double(), ln.bias.double(), ln.eps)


@pytest.mark.parametrize("M,F,kind", [(12537, 1024, "plain"), (4100, 1024, "large"), (4100, 2048, "wide-range"),
                                      (333, 64, "plain"), (4100, 1024, "cancelling")])
def test_fused_ffn_vs_float64(M, F, kind):
    """egtr_ffn_x6_f32 -- fc1 + ReLU + fc2 (+ residual + LayerNorm + position output) in one launch, the hidden activation
    never in HBM -- against float64, and against the fp32 composition it replaces (dd:1335-1345): no further from float64
    than 2.5x that composition (+ a floor of 2e-6 of the output scale), rows of every magnitude."""
    from egtr_amd import ops
    fc1, fc2, ln = _ffn_modules(F, 11 + F)
    rng = W.rng_inputs(6000 + M)
    x = torch.from_numpy(rng.standard_normal((M, 256))).float()
    if kind == "large":
        x = x * 300
    elif kind == "wide-range":
        x = x * torch.pow(2.0, torch.from_numpy(rng.integers(-40, 30, (M, 1))).float())
    elif kind == "cancelling":
        x[:, 1::2] = -x[:, 0::2] * (1 + 1e-6)
        with torch.no_grad():
            fc1.weight[:, 1::2] = fc1.weight[:, 0::2]
    pos = torch.from_numpy(rng.standard_normal((M, 256))).float()
    import copy
    fd1, fd2, lnd = (copy.deepcopy(m).to(DEV) for m in (fc1, fc2, ln))
    xd = x.to(DEV)
    with torch.no_grad():
        y_plain = ops.ffn_fused(xd, fd1, fd2)
        y_ln, y_pos = ops.ffn_fused(xd, fd1, fd2, lnd, pos.to(DEV))
        h32 = torch.relu(torch.nn.functional.linear(xd, fd1.weight, fd1.bias))
        c_plain = torch.nn.functional.linear(h32, fd2.weight, fd2.bias)
        c_ln = lnd(xd + c_plain)
    r_plain, r_ln = _ffn_ref64(x, fc1, fc2), _ffn_ref64(x, fc1, fc2, ln)
    for got, comp, ref in ((y_plain, c_plain, r_plain), (y_ln, c_ln, r_ln)):
        e, e32 = (got.cpu().double() - ref).abs(), (comp.cpu().double() - ref).abs()
        scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)            # per row: rows differ by many binades
        assert torch.isfinite(got).all()
        assert ((e / scale).amax(1) <= 2.5 * (e32 / scale).amax(1) + 2e-6).all(), (kind, float((e / scale).max()))
    assert torch.equal(y_pos, y_ln + pos.to(DEV))
    assert torch.equal(y_ln, ops.ffn_fused(xd, fd1, fd2, lnd))               # bit-reproducible, with or without pos


def test_fused_ffn_non_finite_rows_only():
    """An inf / NaN input element makes exactly ITS output row non-finite; every other row is bit-identical."""
    from egtr_amd import ops
    fc1, fc2, ln = _ffn_modules(1024, 5)
    M = 4100
    x = torch.from_numpy(W.rng_inputs(6100).standard_normal((M, 256))).float()
    bad = {0: float("inf"), 63: float("nan"), 64: float("-inf"), 4099: float("nan")}
    xb = x.clone()
    for r, v in bad.items():
        xb[r, (5 * r) % 256] = v
    import copy
    fd1, fd2, lnd = (copy.deepcopy(m).to(DEV) for m in (fc1, fc2, ln))
    with torch.no_grad():
        clean, dirty = ops.ffn_fused(x.to(DEV), fd1, fd2, lnd).cpu(), ops.ffn_fused(xb.to(DEV), fd1, fd2, lnd).cpu()
    good = torch.ones(M, dtype=torch.bool)
    good[list(bad)] = False
    assert torch.equal(clean[good], dirty[good])
    assert not torch.isfinite(dirty[~good]).any()


@pytest.mark.parametrize("M,kind", [(12537, "plain"), (4100, "large"), (4133, "wide-range"), (65, "plain")])
def test_fused_projection_layernorm_vs_float64(M, kind):
    """egtr_proj_ln_x6_f32 -- the attention output projection + residual + LayerNorm (+ position output) in one launch
    (dd:1102, 1326-1330) -- against float64 and against the fp32 composition it replaces: per row within 2.5x (+ 2e-6 of
    the row's scale)."""
    import copy
    from egtr_amd import ops
    g = torch.Generator().manual_seed(77)
    lin, ln = torch.nn.Linear(256, 256), torch.nn.LayerNorm(256)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(256, 256, generator=g) / 16)
        lin.bias.copy_(torch.randn(256, generator=g) * 0.3)
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
        ln.bias.copy_(0.2 * torch.randn(256, generator=g))
    rng = W.rng_inputs(6300 + M)
    x = torch.from_numpy(rng.standard_normal((M, 256))).float()
    res = torch.from_numpy(rng.standard_normal((M, 256))).float()
    if kind == "large":
        x, res = x * 300, res * 300
    elif kind == "wide-range":
        sc = torch.pow(2.0, torch.from_numpy(rng.integers(-40, 30, (M, 1))).float())
        x, res = x * sc, res * sc
    pos = torch.from_numpy(rng.standard_normal((M, 256))).float()
    lind, lnd = copy.deepcopy(lin).to(DEV), copy.deepcopy(ln).to(DEV)
    with torch.no_grad():
        y_plain = ops.proj_ln_fused(x.to(DEV), lind)
        y_ln, y_pos = ops.proj_ln_fused(x.to(DEV), lind, res.to(DEV), lnd, pos.to(DEV))
        c_plain = lind(x.to(DEV))
        c_ln = lnd(res.to(DEV) + c_plain)
    r_plain = x.double() @ lin.weight.double().t() + lin.bias.double()
    r_ln = torch.nn.functional.layer_norm(res.double() + r_plain, (256,), ln.weight.double(), ln.bias.double(), ln.eps)
    for got, comp, ref in ((y_plain, c_plain, r_plain), (y_ln, c_ln, r_ln)):
        e, e32 = (got.cpu().double() - ref).abs(), (comp.cpu().double() - ref).abs()
        scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
        assert torch.isfinite(got).all()
        assert ((e / scale).amax(1) <= 2.5 * (e32 / scale).amax(1) + 2e-6).all(), (kind, float((e / scale).max()))
    assert torch.equal(y_pos, y_ln + pos.to(DEV))


@pytest.mark.parametrize("M,nw,with_bias", [(12537, 6, False), (4100, 6, True), (65, 1, True), (6000, 3, False)])
def test_fused_multi_projection_vs_float64(M, nw, with_bias):
    """egtr_proj_multi_x6_f32 -- the decoder's cross-attention value projections of the encoder output (dd:1048-1049), all
    layers in one launch -- against float64 and the fp32 vendor product per weight; a non-finite row stays in its row."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(91 + nw)
    w = torch.randn(nw, 256, 256, generator=g) / 16
    b = torch.randn(nw, 256, generator=g) * 0.3 if with_bias else None
    rng = W.rng_inputs(6400 + M)
    x = torch.from_numpy(rng.standard_normal((M, 256))).float() * 3
    x[M // 2, 7] = float("inf")
    with torch.no_grad():
        w_xs = ops.xs_split(w.reshape(nw * 256, 256).to(DEV), weights=True)
        got = ops.proj_multi_fused(x.to(DEV), w_xs, nw, None if b is None else b.reshape(-1).to(DEV)).cpu()
    assert got.shape == (nw, M, 256)
    bad = torch.zeros(M, dtype=torch.bool)
    bad[M // 2] = True
    assert not torch.isfinite(got[:, bad]).all(-1).any()
    assert torch.isfinite(got[:, ~bad]).all()
    xd = x[~bad].double()
    for i in range(nw):
        ref = xd @ w[i].double().t() + (b[i].double() if b is not None else 0)
        comp = (x[~bad].to(DEV) @ w[i].to(DEV).t()).cpu().double() + (b[i].double() if b is not None else 0)
        e, e32 = (got[i][~bad].double() - ref).abs(), (comp - ref).abs()
        scale = ref.abs().amax(1, keepdim=True)
        assert ((e / scale).amax(1) <= 2.5 * (e32 / scale).amax(1) + 2e-6).all(), (i, float((e / scale).max()))


@pytest.mark.parametrize("M,F,kind", [(12537, 1024, "plain"), (4100, 1024, "large"), (4133, 2048, "wide-range"),
                                      (333, 64, "plain"), (65, 128, "non-finite")])
def test_fused_encoder_tail_vs_float64(M, F, kind):
    """egtr_encoder_tail_x6_f32 -- output projection + residual + LayerNorm, then the FFN block + residual + LayerNorm
    (+ position output), ONE launch, the intermediate states never in memory (dd:1102, 1326-1345) -- against float64 and
    against the fp32 composition it replaces: per row no further from float64 than 2.5x that composition (+ 2e-6 of the row
    scale); a non-finite context element makes exactly its own output row non-finite."""
    import copy
    from egtr_amd import ops
    fc1, fc2, ln2 = _ffn_modules(F, 21 + F)
    g = torch.Generator().manual_seed(97)
    proj, ln1 = torch.nn.Linear(256, 256), torch.nn.LayerNorm(256)
    with torch.no_grad():
        proj.weight.copy_(torch.randn(256, 256, generator=g) / 16)
        proj.bias.copy_(torch.randn(256, generator=g) * 0.3)
        ln1.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
        ln1.bias.copy_(0.2 * torch.randn(256, generator=g))
    rng = W.rng_inputs(6600 + M)
    ctx = torch.from_numpy(rng.standard_normal((M, 256))).float()
    hid = torch.from_numpy(rng.standard_normal((M, 256))).float()
    if kind == "large":
        ctx, hid = ctx * 300, hid * 300
    elif kind == "wide-range":
        sc = torch.pow(2.0, torch.from_numpy(rng.integers(-40, 30, (M, 1))).float())
        ctx, hid = ctx * sc, hid * sc
    bad = torch.zeros(M, dtype=torch.bool)
    if kind == "non-finite":
        ctx[7, 3], ctx[40, 200] = float("inf"), float("nan")
        bad[[7, 40]] = True
    pos = torch.from_numpy(rng.standard_normal((M, 256))).float()
    mods = [copy.deepcopy(m).to(DEV) for m in (proj, ln1, fc1, fc2, ln2)]
    with torch.no_grad():
        y, y_pos = ops.encoder_tail_fused(ctx.to(DEV), hid.to(DEV), *mods, pos.to(DEV))
        y_nopos = ops.encoder_tail_fused(ctx.to(DEV), hid.to(DEV), *mods)
        y1c = mods[1](hid.to(DEV) + mods[0](ctx.to(DEV)))
        comp = mods[4](y1c + mods[3](torch.relu(mods[2](y1c))))
    assert torch.equal(y[~bad], y_nopos[~bad]) and torch.equal(y_pos[~bad], (y + pos.to(DEV))[~bad])
    assert not torch.isfinite(y[bad]).all(-1).any() and torch.isfinite(y[~bad]).all()
    y1 = torch.nn.functional.layer_norm(hid.double() + ctx.double() @ proj.weight.double().t() + proj.bias.double(), (256,),
                                        ln1.weight.double(), ln1.bias.double(), ln1.eps)
    ref = _ffn_ref64(y1, fc1, fc2, ln2)
    e, e32 = (y.cpu().double() - ref).abs()[~bad], (comp.cpu().double() - ref).abs()[~bad]
    scale = ref[~bad].abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert ((e / scale).amax(1) <= 2.5 * (e32 / scale).amax(1) + 2e-6).all(), (kind, float((e / scale).max()))
