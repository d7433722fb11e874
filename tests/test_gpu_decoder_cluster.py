"""GPU (-m gpu): the one-launch-per-layer decoder (csrc/dec_layer.hip, egtr_decoder_layer_f32) against the per-operation
decoder of the same model -- which tests/test_gpu_model.py pins to the reference's fixtures (model/deformable_detr.py:
1390-1489, 1774-1968) -- on the same weights and inputs: states of every layer, the retained scaled-q / k maps
(dd:1179-1185), and the heads that read them."""
import numpy as np
import pytest
import torch

import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(num_queries, dec_layers, seed=0):
    from egtr_amd.deformable_detr import DeformableDetrConfig
    from egtr_amd.egtr import DetrForSceneGraphGeneration
    cfg = DeformableDetrConfig(num_queries=num_queries, encoder_layers=1, decoder_layers=dec_layers, dropout=0.1,
                               auxiliary_loss=False)
    for k, v in dict(num_labels=17, num_rel_labels=9, ce_loss_coefficient=2.0, rel_loss_coefficient=15.0,
                     connectivity_loss_coefficient=30.0, smoothing=1e-14, rel_sample_negatives=80,
                     rel_sample_nonmatching=80, rel_sample_negatives_largest=True, rel_sample_nonmatching_largest=True,
                     use_freq_bias=True, use_log_softmax=False, freq_bias_eps=1e-12, logit_adjustment=False,
                     logit_adj_tau=0.3).items():
        setattr(cfg, k, v)
    torch.manual_seed(seed)
    model = DetrForSceneGraphGeneration(cfg, fg_matrix=W.fg_matrix(17, 9)).to(DEV).eval()
    # random-init LayerNorms are the identity and the attention logits' weights are zero: give both some structure
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "layer_norm" in n or "attention_weights" in n:
                p.add_(0.2 * torch.randn(p.shape, generator=g).to(DEV))
    return model


def _inputs(batch, h, w, pad=True, seed=3):
    g = torch.Generator(device="cpu").manual_seed(seed)
    pv = torch.randn(batch, 3, h, w, generator=g)
    pm = torch.ones(batch, h, w, dtype=torch.long)
    if pad and batch > 1:
        pm[1, h - h // 5:, :] = 0
        pm[1, :, w - w // 4:] = 0
        pv[1] = pv[1] * pm[1][None].float()
    return pv.to(DEV), pm.to(DEV)


def _run(model, pv, pm, fused, base=False):
    """Whole model (heads), or with ``base`` the DeformableDetrModel under it (decoder states, retained q / k)."""
    from egtr_amd import decoder_fused
    old = decoder_fused.ENABLED
    decoder_fused.ENABLED = fused
    try:
        with torch.no_grad():
            return (model.model if base else model)(pixel_values=pv, pixel_mask=pm, output_attentions=False,
                                                    output_attention_states=True, output_hidden_states=True)
    finally:
        decoder_fused.ENABLED = old


@pytest.mark.parametrize("num_queries,batch,hw", [(200, 1, (160, 224)), (37, 2, (128, 160)), (300, 2, (96, 128)),
                                                  (8, 3, (96, 128))])
def test_cluster_decoder_matches_the_per_operation_decoder(num_queries, batch, hw, monkeypatch):
    from egtr_amd import decoder_fused, ops
    model = _model(num_queries, 3)
    pv, pm = _inputs(batch, *hw)
    ref, ref_base = _run(model, pv, pm, fused=False), _run(model, pv, pm, fused=False, base=True)
    before = dict(ops.FALLBACKS)
    out, out_base = _run(model, pv, pm, fused=True), _run(model, pv, pm, fused=True, base=True)
    assert ops.FALLBACKS == before, "the cluster kernel was refused"
    assert decoder_fused.ENABLED
    assert decoder_fused.read_status(torch.device(DEV)) == 0
    tol = 2e-4
    a, b = out_base.decoder_hidden_states, ref_base.decoder_hidden_states
    assert len(a) == len(b) == 4
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape
        assert (x - y).abs().max() < tol, ("hidden state", i, float((x - y).abs().max()))
    for name in ("decoder_attention_queries", "decoder_attention_keys"):
        qa, qb = getattr(out_base, name), getattr(ref_base, name)
        assert len(qa) == len(qb) == 3
        for i, (x, y) in enumerate(zip(qa, qb)):
            assert tuple(x.shape) == tuple(y.shape) == (batch, 8, num_queries, 32)
            assert (x - y).abs().max() < tol, (name, i, float((x - y).abs().max()))
    for k in ("logits", "pred_boxes", "pred_rel", "pred_connectivity"):
        assert (out[k] - ref[k]).abs().max() < tol, (k, float((out[k] - ref[k]).abs().max()))


def test_cluster_decoder_is_deterministic_and_survives_graph_replay():
    """The partial sums are added in head order: on the same encoder output two runs of the decoder are bit-identical (the
    vendor convolutions in front of it need not be); a HIP-graph replay of the whole model matches the eager result."""
    from egtr_amd.runtime import GraphedForward
    model = _model(200, 6)
    pv, pm = _inputs(1, 160, 224)
    for _ in range(3):   # several forwards in a row: the seam between two forwards reuses the tagged buffers
        a = _run(model, pv, pm, fused=True, base=True)
    enc = (a.encoder_last_hidden_state,)
    from egtr_amd import decoder_fused
    assert decoder_fused.ENABLED
    with torch.no_grad():
        b = model.model(pixel_values=pv, pixel_mask=pm, encoder_outputs=enc, output_attention_states=True,
                        output_hidden_states=True)
        c = model.model(pixel_values=pv, pixel_mask=pm, encoder_outputs=enc, output_attention_states=True,
                        output_hidden_states=True)
    for x, y in zip(b.decoder_hidden_states, c.decoder_hidden_states):
        assert torch.equal(x, y)
    assert all(torch.equal(x, y) for x, y in zip(b.decoder_attention_queries, c.decoder_attention_queries))
    eager = _run(model, pv, pm, fused=True)
    fwd = GraphedForward(model, enabled=True, strict=True)
    with torch.no_grad():
        for _ in range(3):
            out = fwd(pv, pm)
    torch.cuda.synchronize()
    assert fwd.graphed
    assert (out["pred_rel"] - eager.pred_rel).abs().max() < 1e-5 and (out["logits"] - eager.logits).abs().max() < 1e-5


def test_graph_replays_and_eager_runs_do_not_share_workspace_memory():
    """A captured graph keeps pointers to its barrier counters and partial sums: the buffers must outlive the capture call
    (they were dropped once: every replay then scribbled over whatever eager tensor had received the memory, and one run in
    ten came out wrong).  Replays interleaved with eager forwards on fresh inputs, all compared."""
    from egtr_amd.runtime import GraphedForward
    model = _model(40, 2)
    fwd = GraphedForward(model, enabled=True, strict=True)
    for it in range(25):
        pv, pm = _inputs(1, 160, 224, seed=100 + it)
        a = _run(model, pv, pm, fused=True)
        b = _run(model, pv, pm, fused=True)
        with torch.no_grad():
            r = fwd(pv, pm)
        assert fwd.graphed
        assert (a.pred_rel - b.pred_rel).abs().max() < 1e-5, it
        assert (r["pred_rel"] - a.pred_rel).abs().max() < 1e-5, it


def test_cluster_decoder_rejects_what_it_does_not_serve():
    import ctypes
    from egtr_amd import _lib, decoder_fused
    lib = _lib.lib()
    a = decoder_fused.EgtrDecoderLayer()
    assert lib.egtr_decoder_layer_f32(None, ctypes.byref(a)) == -1
    assert lib.egtr_decoder_layer_f32(None, None) == -1
    model = _model(20, 2)
    pv, pm = _inputs(1, 96, 128)
    model.model.decoder.layers[0].fc1 = torch.nn.Linear(256, 512).to(DEV)   # not the kernel's 1024 hidden units
    model.model.decoder.layers[0].fc2 = torch.nn.Linear(512, 256).to(DEV)
    assert not decoder_fused.supported(model.model.decoder, torch.zeros(1, 20, 256, device=DEV),
                                       torch.zeros(1, 20, 256, device=DEV), torch.zeros(1, 20, 2, device=DEV),
                                       torch.zeros(1, 5, 256, device=DEV), False)
    out = _run(model, pv, pm, fused=True)   # served by the per-operation decoder
    assert torch.isfinite(out.pred_rel).all()


def test_a_refusing_device_falls_back_to_the_per_operation_decoder_and_says_so(monkeypatch):
    """What happens on a device whose dispatch does not keep a cluster on one XCD (or where a barrier times out): the status
    word read after the first eager run is non-zero -> DecoderClusterError -> the per-operation decoder serves the call, the
    fall-off is counted and announced once; under STRICT_FAST_PATH it is an error instead."""
    import warnings
    from egtr_amd import decoder_fused, ops
    model = _model(24, 2)
    pv, pm = _inputs(1, 96, 128)
    ref = _run(model, pv, pm, fused=False)
    monkeypatch.setattr(decoder_fused, "_CHECKED", set())
    orig = decoder_fused._workspace

    def poisoned(dev, key):
        ws = orig(dev, key)
        ws[1].fill_(2)   # "the workgroups of a cluster were spread over several XCDs"
        return ws

    monkeypatch.setattr(decoder_fused, "_workspace", poisoned)
    monkeypatch.setattr(ops, "FALLBACKS", {})
    monkeypatch.setattr(decoder_fused, "ENABLED", True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            out = model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
    assert ops.FALLBACKS.get("decoder_cluster") == 1 and any("decoder_cluster" in str(x.message) for x in w)
    assert not decoder_fused.ENABLED                      # switched off for the process
    assert (out.pred_rel - ref.pred_rel).abs().max() < 1e-5
    monkeypatch.setattr(decoder_fused, "ENABLED", True)
    monkeypatch.setattr(decoder_fused, "_CHECKED", set())
    monkeypatch.setattr(ops, "STRICT_FAST_PATH", True)
    with pytest.raises(ops.FastPathError):
        with torch.no_grad():
            model(pixel_values=pv, pixel_mask=pm, output_attention_states=True)
    monkeypatch.setattr(decoder_fused, "_workspace", orig)
    for ws in decoder_fused._WORKSPACES.values():         # leave clean status words behind
        ws[1].zero_()


def test_a_barrier_time_out_poisons_the_states_and_is_reported_late_without_a_sync():
    """after the first verified run a time-out used to go unnoticed and hand out plausible-looking
    decoder states.  Fault injection (include/egtr_hip_test.h ``egtr_test_decoder_drop_arrival``): one wave of cluster 0
    skips an arrival, its cluster's second barrier times out.  Then (1) the waves that gave up NaN-poison what they hand
    out -- the final states of the affected query rows are NaN, never plausible numbers; (2) the sticky status word is
    picked up by the NEXT ``poll_status`` without a synchronisation in between, for eager workspaces and for the workspaces
    baked into captured graphs alike; (3) after the report the workspace is whole again and results are right."""
    import hip_test_abi
    from egtr_amd import decoder_fused
    from egtr_amd.runtime import GraphedForward
    dev = torch.device(DEV)
    model = _model(24, 2)
    pv, pm = _inputs(1, 96, 128)
    good = _run(model, pv, pm, fused=True, base=True).intermediate_hidden_states.clone()   # [B, Ld, N, 256]; first, verified run
    assert decoder_fused.read_status(dev) == 0 and torch.isfinite(good).all()
    decoder_fused.poll_status(dev, wait=True)            # drain a pending poll of earlier tests
    try:
        hip_test_abi.decoder_drop_arrival(True)
        bad = _run(model, pv, pm, fused=True, base=True).intermediate_hidden_states
        torch.cuda.synchronize()
    finally:
        hip_test_abi.decoder_drop_arrival(False)
    assert torch.isnan(bad[0, 0, :8]).all(), "layer 0, cluster 0 (query rows 0-7) must be NaN-poisoned"
    # (close, not equal: the MIOpen convolutions in front of the encoder are not bit-reproducible run to run)
    assert (bad[0, 0, 8:] - good[0, 0, 8:]).abs().max() < 1e-4, "layer 0: the other clusters passed their barriers, untouched"
    assert torch.isnan(bad[0, 1]).all(), "layer 1 attends over the poisoned keys / values: every row is NaN"
    assert decoder_fused.read_status(dev) & 1
    assert decoder_fused.poll_status(dev) == 0           # queues the copy; nothing to report yet
    with pytest.raises(decoder_fused.DecoderClusterError, match="timed out"):
        decoder_fused.poll_status(dev, wait=True)        # the verdict of the previous poll
    assert decoder_fused.read_status(dev) == 0           # reported once, workspace reset
    again = _run(model, pv, pm, fused=True, base=True).intermediate_hidden_states
    assert (again - good).abs().max() < 1e-4
    # the same through a captured graph: its workspace is not an eager one, GraphedForward polls it every status_every calls
    g = GraphedForward(model, enabled=True, strict=True, status_every=2)
    ok = g(pv, pm).pred_rel.clone()                      # call 1: capture + replay
    try:
        hip_test_abi.decoder_drop_arrival(True)            # (a launch-time argument: replays carry what was captured --
        g.invalidate()                                   #  so re-capture with the fault on)
        r2 = g(pv, pm)                                   # call 2: capture with the fault; poll queued
        torch.cuda.synchronize()
        assert torch.isnan(r2.pred_rel).any()
    finally:
        hip_test_abi.decoder_drop_arrival(False)
    g(pv, pm)                                            # call 3: no poll (status_every = 2)
    with pytest.raises(decoder_fused.DecoderClusterError):
        torch.cuda.synchronize()
        g(pv, pm)                                        # call 4: reads the verdict of call 2's poll
    g.invalidate()
    fine = g(pv, pm).pred_rel
    assert (fine - ok).abs().max() < 1e-5
    assert decoder_fused.read_status(dev) == 0
