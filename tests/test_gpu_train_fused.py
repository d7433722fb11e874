"""GPU (-m gpu): the training form of the encoder layer (egtr_amd.ops.EncoderLayerTrainFunction, round 4) -- one autograd node
per layer whose kernels carry the glue of the per-op composition in their epilogues -- against

  * a float64 statement of the reference layer (model/deformable_detr.py:1283-1358 + :1026-1104 in train mode) under torch
    autograd on the device, with the SAME dropout masks: output, input / position gradients and all sixteen parameter
    gradients;
  * the per-op composition of rounds 2 / 3 (EGTR_ENCODER_TRAIN_FUSED = 0) on the same inputs without dropout;
  * the new kernels on their own: dropout + residual + LayerNorm forward / backward, the split-bf16 GEMM's epilogue options,
    the weight-gradient kernel's position rows / row mask.
"""
import pytest
import torch
import torch.nn.functional as F

import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(75, 125), (38, 63), (19, 32), (10, 16)]      # the 600x1000 pyramid: S = 12 537


def _layer(seed=0, dropout=0.1):
    from egtr_amd.deformable_detr import DeformableDetrConfig, DeformableDetrEncoderLayer
    torch.manual_seed(seed)
    cfg = DeformableDetrConfig(dropout=dropout)
    layer = DeformableDetrEncoderLayer(cfg)
    with torch.no_grad():   # the reference init leaves offsets / attention weights input-independent: perturb them
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.02)
        layer.self_attn.attention_weights.weight.normal_(0, 0.05)
        layer.self_attn.attention_weights.bias.normal_(0, 0.1)
        for ln in (layer.self_attn_layer_norm, layer.final_layer_norm):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.1)
    return layer.to(DEV).train()


def _inputs(B, seed=1, SHAPES=SHAPES):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in SHAPES)
    x = torch.randn(B, S, 256, generator=g)
    pos = torch.randn(B, S, 256, generator=g) * 0.5
    refs = []
    for (h, w) in SHAPES:
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
        refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(B, S, 4, 2).contiguous()
    mask = torch.ones(B, S, dtype=torch.bool)
    if B > 1:   # image 1: the right 20 % of every level is padding
        o = 0
        for (h, w) in SHAPES:
            m = torch.ones(h, w, dtype=torch.bool)
            m[:, int(0.8 * w):] = False
            mask[1, o:o + h * w] = m.reshape(-1)
            o += h * w
    shp = torch.as_tensor(SHAPES, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    gy = torch.randn(B, S, 256, generator=g)
    return [t.to(DEV) for t in (x, pos, ref, mask, shp, lsi, gy)]


def _reference_f64(layer, x, pos, ref, mask, shp, lsi, m1, m2, p):
    """The reference layer in float64 under autograd (MSDA through the f64 HIP entry, everything else torch)."""
    from egtr_amd.ops import MultiScaleDeformableAttentionFunction
    sa = layer.self_attn
    P = {n: q.detach().double().requires_grad_(True) for n, q in layer.named_parameters()}
    x = x.double().requires_grad_(True)
    pos = pos.double().requires_grad_(True)
    B, S, _ = x.shape
    scale = 1.0 / (1.0 - p) if p > 0 else 1.0
    value = F.linear(x, P["self_attn.value_proj.weight"], P["self_attn.value_proj.bias"])
    value = value.masked_fill(~mask[..., None], 0.0).view(B, S, 8, 32)
    q = x + pos
    off = F.linear(q, P["self_attn.sampling_offsets.weight"], P["self_attn.sampling_offsets.bias"]).view(B, S, 8, 4, 4, 2)
    aw = F.softmax(F.linear(q, P["self_attn.attention_weights.weight"], P["self_attn.attention_weights.bias"])
                   .view(B, S, 8, 16), -1).view(B, S, 8, 4, 4)
    norm = torch.stack([shp[..., 1], shp[..., 0]], -1).double()
    loc = ref.double()[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    att = MultiScaleDeformableAttentionFunction.apply(value.contiguous(), shp, lsi, loc.contiguous(), aw.contiguous(), 64)
    a = F.linear(att, P["self_attn.output_proj.weight"], P["self_attn.output_proj.bias"])
    if m1 is not None:
        a = a * m1.view(B, S, 256).double() * scale
    y1 = F.layer_norm(x + a, (256,), P["self_attn_layer_norm.weight"], P["self_attn_layer_norm.bias"], layer.self_attn_layer_norm.eps)
    h = F.relu(F.linear(y1, P["fc1.weight"], P["fc1.bias"]))
    f = F.linear(h, P["fc2.weight"], P["fc2.bias"])
    if m2 is not None:
        f = f * m2.view(B, S, 256).double() * scale
    y2 = F.layer_norm(y1 + f, (256,), P["final_layer_norm.weight"], P["final_layer_norm.bias"], layer.final_layer_norm.eps)
    return y2, x, pos, P


@pytest.mark.parametrize("B,p", [(2, 0.1), (1, 0.0)])
def test_encoder_layer_train_node_vs_float64_reference(B, p):
    from egtr_amd import ops
    layer = _layer(dropout=p)
    x, pos, ref, mask, shp, lsi, gy = _inputs(B)
    S = x.shape[1]
    masks = None
    if p > 0:
        g = torch.Generator().manual_seed(3)
        mm = (torch.rand(2, B * S, 256, generator=g) < 1.0 - p).to(torch.uint8).to(DEV)
        masks = (mm[0], mm[1])
    xg, pg = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    assert ops.encoder_layer_train_supported(layer, xg, pg, ref, mask, False)
    y = ops.encoder_layer_train(layer, xg, mask, pg, ref, shp, lsi, masks=masks)
    y.backward(gy)
    y_ref, x64, p64, P = _reference_f64(layer, x, pos, ref, mask, shp, lsi, masks[0] if masks else None,
                                        masks[1] if masks else None, p)
    y_ref.backward(gy.double())

    def close(got, want, rel, name):
        want = want.float()
        err = (got.float() - want).abs().max().item()
        assert err <= rel * max(1.0, want.abs().max().item()), (name, err, want.abs().max().item())

    def close_rows(got, want, rel, name, max_bad=0.02):
        """Bilinear sampling is continuous but NOT differentiable where a sample crosses a pixel boundary: fp32 and fp64
        place a handful of the 6.4 M sample coordinates on different sides (coordinate rounding ~1e-5 px), and the location
        gradient of exactly those queries differs at O(1).  Rows are therefore judged one by one: all but a small fraction
        within `rel`, and the layer-level aggregate (Frobenius norm) within 1 %."""
        want = want.float().reshape(-1, want.shape[-1])
        err = (got.float().reshape(want.shape) - want).abs().max(1)[0]
        bad = (err > rel * max(1.0, want.abs().max().item())).float().mean().item()
        frob = ((got.float().reshape(want.shape) - want).norm() / want.norm()).item()
        print(f"{name}: rows outside {rel:g} relative: {100 * bad:.3f} %, relative Frobenius error {frob:.2e}")
        assert bad <= max_bad and frob < 1e-2, (name, bad, frob)

    close(y.detach(), y_ref.detach(), 2e-5, "output")
    close_rows(xg.grad, x64.grad, 1e-4, "grad x")
    close_rows(pg.grad, p64.grad, 1e-4, "grad pos")
    report = {}
    for n, q in layer.named_parameters():
        assert q.grad is not None, n
        w = P[n].grad.float()
        report[n] = ((q.grad - w).abs().max().item(), w.abs().max().item(), ((q.grad - w).norm() / w.norm()).item())
    print({n: f"max {e:.2e} / {m:.2e}, frobenius {f:.1e}" for n, (e, m, f) in report.items()})
    for n, (e, m, f) in report.items():
        # Parameter gradients sum over all rows.  Two kinds of non-differentiable points put isolated O(1) terms on different
        # sides in fp32 and fp64: a sample on a pixel boundary (gradients through the sampling locations) and a ReLU input
        # within rounding of zero (ONE such element moved fc1's gradient by 1 % of its largest entry in the first case here).
        # So: the aggregate must agree to fp32 level, single entries to a few percent of the largest.
        assert f < 5e-3 and e <= 3e-2 * max(1.0, m), (n, e, m, f)
    exact = [n for n, (e, m, f) in report.items() if f < 2e-5]
    assert len(exact) >= 4, report      # the gradients behind the last LayerNorm see none of those points: fp32-exact


@pytest.mark.parametrize("B,shapes", [(2, SHAPES), (4, [(100, 167), (50, 84), (25, 42), (13, 21)])])
def test_encoder_layer_train_node_equals_the_per_op_composition(monkeypatch, B, shapes):
    """Same layer, same inputs, dropout 0: the fused node against the round-2 / 3 composition (TokenLinearFunction,
    AddLayerNormFunction, MSDAGeometryFunction, clamp_nonfinite_) -- both fp32, the same split-bf16 products.  Second case:
    the reference's largest training geometry (800x1333, bs 4: 88 892 token rows -- more than 2 048 32-row blocks of
    bias-gradient partials, the other branch of their final reduction)."""
    from egtr_amd import ops
    SHAPES = shapes
    layer = _layer(dropout=0.0)
    x, pos, ref, mask, shp, lsi, gy = _inputs(B, SHAPES=shapes)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "ENCODER_TRAIN_FUSED", fused)
        layer.zero_grad(set_to_none=True)
        xg, pg = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        out = layer(xg, mask, position_embeddings=pg, reference_points=ref, spatial_shapes=shp, level_start_index=lsi,
                    spatial_shapes_list=SHAPES)[0]
        out.backward(gy)
        res.append((out.detach(), xg.grad, pg.grad, {n: q.grad.clone() for n, q in layer.named_parameters()}))
    (y0, gx0, gp0, P0), (y1, gx1, gp1, P1) = res
    assert (y0 - y1).abs().max() < 2e-5
    assert (gx0 - gx1).abs().max() < 1e-4 * max(1.0, float(gx1.abs().max()))
    assert (gp0 - gp1).abs().max() < 1e-4 * max(1.0, float(gp1.abs().max()))
    for n in P0:
        assert (P0[n] - P1[n]).abs().max() < 2e-4 * max(1.0, float(P1[n].abs().max())), n


def test_encoder_layer_train_node_clamps_like_the_reference_when_states_are_non_finite():
    """dd:1346-1351: iff the layer output holds an inf / nan it is clamped to +-(max - 1000) (NaN stays NaN) and clamped
    elements pass no gradient; the decision is a device flag raised by the closing LayerNorm kernel."""
    from egtr_amd import ops
    layer = _layer(dropout=0.0)
    x, pos, ref, mask, shp, lsi, gy = _inputs(1)
    x[0, 7, 3] = float("inf")          # row 7: x + a = inf -> LayerNorm row NaN
    xg, pg = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    y = ops.encoder_layer_train(layer, xg, mask, pg, ref, shp, lsi)
    assert torch.isnan(y[0, 7]).all()
    bad = ~torch.isfinite(y).all(-1)[0]
    y.backward(gy)
    torch.cuda.synchronize()
    # rows that stayed finite (nearly all: the inf reaches other rows only through sampled values) keep finite gradients
    assert int(bad.sum()) < 2000
    assert torch.isfinite(layer.final_layer_norm.bias.grad).all()       # d beta = sum of the MASKED incoming gradient
    want = gy[0][~bad].sum(0)
    assert (layer.final_layer_norm.bias.grad - want).abs().max() < 1e-2


def test_dropout_add_layernorm_kernels_vs_torch():
    from egtr_amd import ops
    g = torch.Generator().manual_seed(11)
    rows = 20000
    x, r, gy = (torch.randn(rows, 256, generator=g).to(DEV) for _ in range(3))
    keep = (torch.rand(rows, 256, generator=g) < 0.9).to(torch.uint8).to(DEV)
    w = (torch.rand(256, generator=g) + 0.5).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    for kp, scale in ((keep, 1.0 / 0.9), (None, 1.0)):
        x64 = x.double().requires_grad_(True)
        r64 = r.double().requires_grad_(True)
        w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
        d = x64 * kp.double() * scale if kp is not None else x64
        y64 = F.layer_norm(r64 + d, (256,), w64, b64, 1e-5)
        y64.backward(gy.double())
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        y = ops.dropout_add_layernorm(x, r, kp, scale, w, b, 1e-5, flag=flag)
        assert int(flag) == 0 and (y.double() - y64).abs().max() < 1e-5
        gs, gx, gbb = ops.dropout_add_layernorm_backward(x, r, kp, scale, w, 1e-5, gy)
        assert (gs.double() - r64.grad).abs().max() < 2e-5
        assert (gx.double() - x64.grad).abs().max() < 2e-5
        assert (gbb[:256].double() - w64.grad).abs().max() < 1e-3 * float(w64.grad.abs().max())
        assert (gbb[256:512].double() - b64.grad).abs().max() < 1e-3 * float(b64.grad.abs().max())
        assert (gbb[512:].double() - x64.grad.sum(0)).abs().max() < 1e-3 * float(x64.grad.sum(0).abs().max())
    xn = x.clone()
    xn[5, 0] = float("nan")
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.dropout_add_layernorm(xn, r, None, 1.0, w, b, 1e-5, flag=flag)
    assert int(flag) == 1


def test_split_gemm_epilogue_options_and_wgrad_options_vs_float64():
    """egtr_linear_split_bf16_ex_f32: row mask, ReLU-backward mask, two addends (one aliasing the output), 32-row column
    partials; egtr_linear_split_bf16_wgrad_ex_f32: position rows added on load, row mask -- against float64."""
    from egtr_amd import ops
    rng = W.rng_inputs(77)
    M, K, N = 4999, 256, 384
    x = torch.from_numpy(rng.standard_normal((M, K))).float().to(DEV)
    w = (torch.from_numpy(rng.standard_normal((N, K))).float() / 16).to(DEV)
    b = torch.from_numpy(rng.standard_normal(N)).float().to(DEV)
    pos = torch.from_numpy(rng.standard_normal((M, K))).float().to(DEV)
    keep = torch.from_numpy((rng.random(M) < 0.8)).to(torch.uint8).to(DEV)
    ref = torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV)
    a1 = torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV)
    a2 = torch.from_numpy(rng.standard_normal((M, N))).float().to(DEV)
    wt = ops.gemm_split_weights(w)
    base = ((x + pos).double() @ w.double().t() + b.double())
    want = torch.where(keep.bool()[:, None], base, torch.zeros_like(base))
    want = torch.where(ref.double() > 0, want, torch.zeros_like(want)) + a1.double() + a2.double()
    out = a1.clone()           # add1 aliases the output
    guard = torch.full(((M + 31) // 32 + 4, N), 7.0, device=DEV)      # 4 canary rows behind the [ceil(M / 32), N] partials
    colp = guard[:(M + 31) // 32]
    y = ops.linear_split_ex([dict(x=x, wt=wt, N=N, b=b, pos=pos, row_keep=keep, relu_ref=ref, add1=out, add2=a2, out=out,
                                  colpart=colp)], M, K)[0]
    assert bool((guard[(M + 31) // 32:] == 7.0).all()), "column partials written beyond ceil(M / 32) rows"
    assert y.data_ptr() == out.data_ptr()
    assert (y.double() - want).abs().max() < 2e-4
    assert (colp.double().sum(0) - want.sum(0)).abs().max() < 2e-3 * float(want.sum(0).abs().max())
    blk = want[:32 * (M // 32)].view(M // 32, 32, N).sum(1)
    assert (colp[:M // 32].double() - blk).abs().max() < 1e-3 * float(blk.abs().max())
    # weight gradient with position rows and a row mask (N, K multiples of 128)
    gmat = torch.from_numpy(rng.standard_normal((M, 128))).float().to(DEV)
    gw = ops._wgrad_ex(gmat, x, x_pos=pos, row_keep=keep)
    want_w = (gmat.double() * keep.double()[:, None]).t() @ (x + pos).double()
    assert (gw.double() - want_w).abs().max() < 2e-5 * float(want_w.abs().max()) + 1e-3
    gw0 = ops._wgrad_ex(gmat, x)
    assert torch.equal(gw0, ops.linear_split_bf16_wgrad(gmat, x))


def test_decoder_value_projection_node_vs_float64():
    """ops.DecoderValueProjTrainFunction: values_l = mask(enc W_l^T + b_l) for six layers, and through arbitrary upstream
    gradients (one layer's output unused: its gradient is None) d enc, d W_l, d b_l -- against float64 autograd."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(21)
    B, S, nl = 2, 4500, 6
    enc = torch.randn(B, S, 256, generator=g).to(DEV)
    mask = (torch.rand(B, S, generator=g) < 0.85).to(DEV)
    lins = [torch.nn.Linear(256, 256).to(DEV) for _ in range(nl)]

    class _L:   # the attribute path the decoder layers expose
        def __init__(self, lin):
            self.encoder_attn = type("A", (), {"value_proj": lin})()

    layers = [_L(m) for m in lins]
    e = enc.clone().requires_grad_(True)
    assert ops.decoder_values_train_supported(e, mask, layers)
    vals = ops.decoder_values_train(e, mask, layers)
    gys = [torch.randn(B, S, 256, generator=g).to(DEV) for _ in range(nl)]
    used = [0, 1, 2, 4, 5]                  # layer 3's values take no part in the loss
    sum((vals[i] * gys[i]).sum() for i in used).backward()
    e64 = enc.double().requires_grad_(True)
    P = [(m.weight.detach().double().requires_grad_(True), m.bias.detach().double().requires_grad_(True)) for m in lins]
    v64 = [F.linear(e64, w, b).masked_fill(~mask[..., None], 0.0) for w, b in P]
    sum((v64[i] * gys[i].double()).sum() for i in used).backward()
    for i in range(nl):
        assert (vals[i].double() - v64[i]).abs().max() < 2e-5, i
    assert (e.grad.double() - e64.grad).abs().max() < 2e-4 * float(e64.grad.abs().max())
    for i in used:
        assert (lins[i].weight.grad.double() - P[i][0].grad).abs().max() < 2e-5 * float(P[i][0].grad.abs().max()) + 1e-3, i
        assert (lins[i].bias.grad.double() - P[i][1].grad).abs().max() < 2e-5 * float(P[i][1].grad.abs().max()) + 1e-3, i
    assert lins[3].weight.grad is None and lins[3].bias.grad is None


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_dropout_add_layer_norm_function_vs_torch(p):
    from egtr_amd import ops
    g = torch.Generator().manual_seed(31)
    x, r, gy = (torch.randn(4, 200, 256, generator=g).to(DEV) for _ in range(3))
    ln = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_(0, 0.1)
    keep = (torch.rand(800, 256, generator=g) < 1 - p).to(torch.uint8).to(DEV) if p > 0 else None
    xg, rg = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    y = ops.dropout_add_layer_norm(xg, rg, ln, p, True, keep=keep)
    y.backward(gy)
    x64, r64 = x.double().requires_grad_(True), r.double().requires_grad_(True)
    w64, b64 = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    d = x64 * keep.view(4, 200, 256).double() / (1 - p) if keep is not None else x64
    y64 = F.layer_norm(r64 + d, (256,), w64, b64, ln.eps)
    y64.backward(gy.double())
    assert (y.double() - y64).abs().max() < 1e-5
    assert (xg.grad.double() - x64.grad).abs().max() < 2e-5 and (rg.grad.double() - r64.grad).abs().max() < 2e-5
    assert (ln.weight.grad.double() - w64.grad).abs().max() < 1e-4 * float(w64.grad.abs().max())
    assert (ln.bias.grad.double() - b64.grad).abs().max() < 1e-4 * float(b64.grad.abs().max())
    # eval mode / no_grad: the plain composition (no dropout)
    with torch.no_grad():
        y_eval = ops.dropout_add_layer_norm(x, r, ln, p, False)
    assert (y_eval - ln(r + x)).abs().max() < 1e-5


def test_multi_weight_tiling_equals_the_single_weight_entries():
    """egtr_gemm_split_tile_weights_multi_f32: several weights (one of them a virtual row-wise concatenation of two) tiled in
    one launch -- bit-identical to tiling each (materialised) weight with the pair entry."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(41)
    w_a = torch.randn(256, 256, generator=g).to(DEV)
    w_b = torch.randn(128, 256, generator=g).to(DEV)
    w_c = torch.randn(1024, 256, generator=g).to(DEV)
    w_d = torch.randn(256, 1024, generator=g).to(DEV)[:, :]
    got = ops.gemm_split_tile_pairs([w_a, (w_a, w_b), w_c, w_d])
    want = [ops.gemm_split_tile_pair(w) for w in (w_a, torch.cat([w_a, w_b], 0), w_c, w_d)]
    for (g0, g1), (w0, w1) in zip(got, want):
        assert g0.shape == w0.shape and g1.shape == w1.shape
        assert torch.equal(g0.view(torch.int16), w0.view(torch.int16)) and torch.equal(g1.view(torch.int16), w1.view(torch.int16))


def test_level_geometry_train_function_gradient_of_level_embed():
    """ops.level_geometry_train: same outputs as the inference kernel, and d level_embed[l] = sum of d pos over the tokens of
    level l (all images) -- against autograd of the explicit `pos + level_embed[level]` composition."""
    from egtr_amd import ops
    g = torch.Generator().manual_seed(51)
    B, H, Wd = 2, 150, 200
    shapes = [(19, 25), (10, 13), (5, 7), (3, 4)]     # S = 652: B * S * 4 bytes is not a multiple of 16 for B = 1, 3, ...
    pm = torch.ones(B, H, Wd, dtype=torch.long)
    pm[1, 120:, :] = 0
    pm[1, :, 170:] = 0
    pm = pm.to(DEV)
    le = torch.randn(4, 256, generator=g).to(DEV).requires_grad_(True)
    mask, pos, vr, ref = ops.level_geometry_train(pm, shapes, le, 128, 10000, 2 * 3.141592653589793)
    with torch.no_grad():
        m0, p0, v0, r0, _bits = ops.level_geometry(pm, shapes, le.detach(), 128, 10000, 2 * 3.141592653589793)
    assert torch.equal(mask, m0) and torch.equal(pos.detach(), p0) and torch.equal(vr, v0) and torch.equal(ref, r0)
    assert pos.requires_grad and not vr.requires_grad and not ref.requires_grad
    gy = torch.randn(pos.shape, generator=g).to(DEV)
    pos.backward(gy)
    sizes = [h * w for h, w in shapes]
    want, o = [], 0
    for n in sizes:
        want.append(gy[:, o:o + n].double().sum((0, 1)))
        o += n
    want = torch.stack(want).float()
    assert (le.grad - want).abs().max() < 1e-3 * float(want.abs().max())
    # an odd number of token rows (the per-level row weights must each be 16-byte aligned on their own)
    le1 = le.detach().clone().requires_grad_(True)
    shapes1 = [(19, 25), (10, 13), (5, 7), (3, 3)]
    _, pos1, _, _ = ops.level_geometry_train(pm[:1], shapes1, le1, 128, 10000, 2 * 3.141592653589793)
    assert pos1.shape[1] % 4 != 0
    pos1.sum().backward()
    want1 = torch.tensor([float(h * w) for h, w in shapes1], device=DEV)[:, None].expand(4, 256)
    assert (le1.grad - want1).abs().max() < 1e-3


# ---- decoder layer training node (round 6) ------------------------------------------------------------------------------------
def _decoder_layer(seed=0, dropout=0.1):
    from egtr_amd.deformable_detr import DeformableDetrConfig, DeformableDetrDecoderLayer
    torch.manual_seed(seed)
    layer = DeformableDetrDecoderLayer(DeformableDetrConfig(dropout=dropout))
    with torch.no_grad():
        layer.encoder_attn.sampling_offsets.weight.normal_(0, 0.02)
        layer.encoder_attn.attention_weights.weight.normal_(0, 0.05)
        layer.encoder_attn.attention_weights.bias.normal_(0, 0.1)
        for ln in (layer.self_attn_layer_norm, layer.encoder_attn_layer_norm, layer.final_layer_norm):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.1)
    return layer.to(DEV).train()


@pytest.mark.parametrize("B,N,p", [(4, 200, 0.1), (2, 37, 0.0), (1, 300, 0.1)])
def test_decoder_layer_train_node_equals_the_per_op_composition(monkeypatch, B, N, p):
    """VERDICT r4 / r5 (asked twice): one autograd node per decoder layer (ops.DecoderLayerTrainFunction; reference layer
    model/deformable_detr.py:1390-1489) against the per-operation composition of the SAME kernels (SkinnyLinearFunction,
    DecoderSelfAttentionFunction, MSDAGeometryFunction, the MSDA Function, DropoutAddLayerNormFunction -- which
    tests/test_gpu_model.py pins to the reference's train fixtures) with the SAME dropout masks: output states, the retained
    scaled-q / k maps, and every gradient -- states, position rows (a batch EXPANSION of the query table, as in the model),
    reference points, the value projection, all 26 parameters -- with an upstream gradient on the states AND on both maps (the
    relation head's route).  Both sides are exact-f32 MFMA; only the summation order at the gradient meeting points differs."""
    from egtr_amd import ops
    shapes = [(19, 32), (10, 16), (5, 8), (3, 4)]
    S = sum(h * w for h, w in shapes)
    layer = _decoder_layer(dropout=p)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, 256, generator=g).to(DEV)
    table = (torch.randn(N, 256, generator=g) * 0.5).to(DEV)
    refp = torch.rand(B, N, 1, 2, generator=g).to(DEV)
    vr = (0.7 + 0.3 * torch.rand(B, 1, 4, 2, generator=g)).to(DEV)
    value = torch.randn(B, S, 256, generator=g).to(DEV)
    shp = torch.as_tensor(shapes, dtype=torch.long, device=DEV)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    gy, gq, gk = (torch.randn(B, N, 256, generator=g).to(DEV) for _ in range(3))
    masks = torch.empty(3, B * N, 256, dtype=torch.uint8, device=DEV).bernoulli_(1.0 - p) if p > 0 else None
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "DECODER_TRAIN_FUSED", fused)
        monkeypatch.setattr(ops, "FALLBACKS", {})
        layer.zero_grad(set_to_none=True)
        xg, tg = x.clone().requires_grad_(True), table.clone().requires_grad_(True)
        rg, vg = refp.clone().requires_grad_(True), value.clone().requires_grad_(True)
        pos = tg.unsqueeze(0).expand(B, -1, -1)
        ref_in = rg * vr                                  # reference_points[:, :, None] * valid_ratios[:, None]
        outs = layer(xg, position_embeddings=pos, reference_points=ref_in, spatial_shapes=shp, level_start_index=lsi,
                     encoder_hidden_states=value, encoder_attention_mask=None, output_attention_states=True,
                     spatial_shapes_list=shapes, precomputed_value=vg, dropout_masks=masks)
        y, qm, km = outs[0], outs[1], outs[2]
        assert tuple(qm.shape) == (B, 8, N, 32)
        gqm = gq.view(B, N, 8, 32).transpose(1, 2)
        gkm = gk.view(B, N, 8, 32).transpose(1, 2)
        torch.autograd.backward([y, qm, km], [gy, gqm, gkm])
        res.append(dict(y=y.detach(), q=qm.detach().clone(), k=km.detach().clone(), gx=xg.grad, gt=tg.grad, gr=rg.grad,
                        gv=vg.grad, P={n: (q_.grad.clone() if q_.grad is not None else None)
                                       for n, q_ in layer.named_parameters()}))
        assert not ops.FALLBACKS, ops.FALLBACKS
    a, b = res
    assert (a["y"] - b["y"]).abs().max() < 2e-5
    assert (a["q"] - b["q"]).abs().max() < 1e-5 and (a["k"] - b["k"]).abs().max() < 1e-5
    for key in ("gx", "gt", "gr", "gv"):
        assert a[key] is not None and b[key] is not None, key
        assert (a[key] - b[key]).abs().max() < 2e-4 * max(1.0, float(b[key].abs().max())), key
    assert len(a["P"]) == 26
    with_grad = [n for n in a["P"] if b["P"][n] is not None]
    assert len(with_grad) == 24          # (the layer's own value_proj is not used here: the values are handed in)
    for n in a["P"]:
        if b["P"][n] is None:
            assert a["P"][n] is None and "value_proj" in n, n
            continue
        assert (a["P"][n] - b["P"][n]).abs().max() < 2e-4 * max(1.0, float(b["P"][n].abs().max())), n
