"""Host side of ``egtr_decoder_layer_f32`` (csrc/dec_layer.hip): the decoder stack at inference as ONE launch per layer.

Reference: model/deformable_detr.py:1774-1968 (the loop), :1390-1489 (the layer).  The kernel runs a layer as clusters of
eight workgroups that meet in the L2 of one XCD; it relies on workgroup ids being dealt round-robin to the XCDs and CHECKS
that on every launch (status word, bit 1).  ``run`` reads the status after the first eager call on a device and raises
``DecoderClusterError`` if a cluster timed out or was spread over XCDs -- the caller then uses the per-operation path and
says so once (``ops.note_fallback``).  After that the status words stay STICKY and are polled without a synchronisation
(``poll_status``: every 64 eager forwards here, every ``status_every`` replays in ``runtime.GraphedForward``, at the end of
``runtime.calculate_fps``); the words of workspaces baked into captured graphs are included.  A wave that gives up on a
barrier NaN-poisons what it hands out (csrc/dec_layer.hip ``reduce_ln``), so a time-out can not yield plausible states.
"""
import ctypes
import os

import torch

from . import _lib
from .load_custom import _stream

# "0": the per-operation decoder (eight launches per layer) -- the A/B switch of tools/forward_breakdown.py
ENABLED = os.environ.get("EGTR_DECODER_CLUSTER", "1") != "0"
MAX_QUERIES = 320
# (Round 5 also carried a barrier-free hand-over of the cluster's partial results, EGTR_DECODER_DATAFLOW: 34 vs 29 us per
# layer -- removed in round 6 together with its switch)


class DecoderClusterError(RuntimeError):
    pass


_P = ctypes.c_void_p


class EgtrDecoderLayer(ctypes.Structure):
    """include/egtr_hip.h: struct EgtrDecoderLayer, field for field."""
    _fields_ = [(n, _P) for n in (
        "x_in", "pos", "q", "k", "v", "reference_points", "valid_ratios", "value", "value_bias", "keep_bits", "spatial_shapes",
        "level_start_index", "x_out", "q_next", "k_next", "v_next", "w_attn_out", "b_attn_out", "ln1_gamma", "ln1_beta",
        "w_off_logit", "b_off_logit", "w_cross_out", "b_cross_out", "ln2_gamma", "ln2_beta", "w_fc1", "b_fc1", "w_fc2",
        "b_fc2", "ln3_gamma", "ln3_beta", "w_qkv_next", "b_qkv_next", "partials", "barriers", "status", "xcc_ids")] + [
        ("q_scale", ctypes.c_float), ("ln_eps", ctypes.c_float), ("batch", ctypes.c_int), ("num_query", ctypes.c_int),
        ("spatial_size", ctypes.c_int), ("x_rows", ctypes.c_int), ("pos_rows", ctypes.c_int), ("qkv_rows", ctypes.c_int),
        ("num_clusters", ctypes.c_int), ("generation", ctypes.c_int), ("ref_rows", ctypes.c_int)]


def pack(w):
    """[N, K] -> [N / 64 tiles][K / 4][64 output columns][4 consecutive k] (N % 64 == 0, K % 4 == 0): the order in which a
    wave of the kernel consumes a weight tile -- one coalesced 16-byte load per lane per four k."""
    n, k = w.shape
    assert n % 64 == 0 and k % 4 == 0
    return w.reshape(n // 64, 64, k // 4, 4).permute(0, 2, 1, 3).contiguous()


def _qkv_pack(attn):
    """Per head two tiles: [q_h | k_h] and [v_h | 0]; biases [8][128] = (q_h, k_h, v_h, 0)."""
    wq, wk, wv = attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight
    z = wq.new_zeros(32, wq.shape[1])
    tiles, bias = [], []
    for h in range(8):
        s = slice(32 * h, 32 * h + 32)
        tiles.append(pack(torch.cat([wq[s], wk[s]], 0)))
        tiles.append(pack(torch.cat([wv[s], z], 0)))
        bias.append(torch.cat([attn.q_proj.bias[s], attn.k_proj.bias[s], attn.v_proj.bias[s], wq.new_zeros(32)]))
    return torch.cat(tiles, 0).contiguous(), torch.cat(bias).contiguous()


def _layer_constants(layer):
    """The packed weights of one decoder layer (cached on the layer, rebuilt when a parameter changes)."""
    from . import ops
    sa, ca = layer.self_attn, layer.encoder_attn
    srcs = [sa.out_proj.weight, ca.sampling_offsets.weight, ca.attention_weights.weight, ca.sampling_offsets.bias,
            ca.attention_weights.bias, ca.output_proj.weight, layer.fc1.weight, layer.fc2.weight, sa.q_proj.weight,
            sa.k_proj.weight, sa.v_proj.weight, sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias]

    def build():
        so, aw = ca.sampling_offsets, ca.attention_weights
        z = so.weight.new_zeros(16, so.weight.shape[1])
        wol = torch.cat([pack(torch.cat([so.weight[32 * h:32 * h + 32], aw.weight[16 * h:16 * h + 16], z], 0))
                         for h in range(8)], 0).contiguous()
        bol = torch.cat([torch.cat([so.bias[32 * h:32 * h + 32], aw.bias[16 * h:16 * h + 16], so.bias.new_zeros(16)])
                         for h in range(8)]).contiguous()
        wqkv, bqkv = _qkv_pack(sa)
        return dict(w_attn_out=pack(sa.out_proj.weight), w_off_logit=wol, b_off_logit=bol,
                    w_cross_out=pack(ca.output_proj.weight), w_fc1=pack(layer.fc1.weight), w_fc2=pack(layer.fc2.weight),
                    w_qkv=wqkv, b_qkv=bqkv)

    return ops.cached_weights(layer, "decoder_cluster_pack", srcs, build)


def supported(decoder, hidden_states, position_embeddings, reference_points, encoder_hidden_states, output_attentions):
    """The configuration the kernel is written for: EGTR's decoder (d_model 256, 8 heads, 4 levels x 4 points, 1024 hidden
    units, ReLU, no box refinement, 2-d reference points) at inference in fp32."""
    import torch.nn.functional as F
    from . import ops
    eligible = (ENABLED and torch.is_tensor(hidden_states) and hidden_states.is_cuda and hidden_states.dtype == torch.float32
                and not torch.is_grad_enabled() and encoder_hidden_states is not None and position_embeddings is not None
                and not output_attentions and decoder.bbox_embed is None and reference_points.shape[-1] == 2)
    if not eligible:   # training, bf16, box refinement, attention maps requested, the switch: other paths by design
        return False
    ok = hidden_states.shape[-1] == 256 and hidden_states.shape[1] <= MAX_QUERIES
    for l in decoder.layers:
        ca, sa = l.encoder_attn, l.self_attn
        ok = ok and (l.activation_fn is F.relu and sa.num_heads == 8 and sa.embed_dim == 256 and ca.n_heads == 8
                     and ca.n_levels == 4 and ca.n_points == 4 and ca.d_model == 256 and l.fc1.out_features == 1024
                     and l.self_attn_layer_norm.eps == l.encoder_attn_layer_norm.eps == l.final_layer_norm.eps
                     and sa.q_proj.bias is not None and l.fc1.bias is not None)
    return ops._gate("decoder_cluster", True, ok,
                     lambda: f"{hidden_states.shape[1]} queries x {hidden_states.shape[-1]} channels: the cluster kernel serves "
                             "d_model 256, 8 heads, 4 x 4 sampling points, 1024 hidden units, ReLU, <= 320 queries")


_WORKSPACES = {}   # (device index, stream handle, shape key) -> persistent buffers of eager launches
_POOL = {}         # (device index, shape key) -> zeroed buffer sets, one per captured graph
_GRAPH_WORKSPACES = []   # buffer sets baked into captured graphs, kept alive for the life of the process
_CHECKED = set()   # device indices whose first eager run was verified
_EAGER_CALLS = {}  # device index -> eager forwards since import (status poll every POLL_EVERY)
POLL_EVERY = 64


def _new_workspace(dev, shape_key):
    """(barriers, status, partials, xcc ids): zeroed ONCE.  The barrier counters only grow."""
    B, N, _ = shape_key
    lib = _lib.lib()
    pf, bw, iw = ctypes.c_longlong(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.egtr_decoder_layer_workspace(B, N, ctypes.byref(pf), ctypes.byref(bw), ctypes.byref(iw)),
               "egtr_decoder_layer_workspace")
    return (torch.zeros(bw.value, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
            torch.zeros(pf.value, dtype=torch.float32, device=dev), torch.empty(iw.value, dtype=torch.int32, device=dev))


def _workspace(dev, shape_key):
    """Launches that share a buffer set must be stream-ordered and of one shape (batch, queries, layers): then the barrier
    counters stay whole.  Eager launches: one set per
    (device, stream, shape).  Under stream capture: a set of its own for the graph being captured, taken from a pool that
    the first eager run of that shape filled (an allocation inside the capture would put memset nodes into the graph; it
    still works and is what happens when the pool is empty)."""
    if torch.cuda.is_current_stream_capturing():
        pool = _POOL.get((dev.index, shape_key))
        ws = pool.pop() if pool else _new_workspace(dev, shape_key)
        # the captured launches keep the POINTERS: the tensors must outlive the graph.  (Dropping them handed the memory back
        # to the allocator, and every replay then wrote its partial sums and barrier counters into whatever eager tensor had
        # been given it since -- tests/test_gpu_model.py::test_graph_replay_matches_eager failed one run in ten.)
        _GRAPH_WORKSPACES.append(ws)
        return ws
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, shape_key)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = _new_workspace(dev, shape_key)
        if (dev.index, shape_key) not in _POOL:
            _POOL[(dev.index, shape_key)] = [_new_workspace(dev, shape_key) for _ in range(4)]
    return ws


def _rows(t, n_rows_per_image):
    """[B, N, 256] (possibly a stride-0 expansion of [N, 256]) -> (dense 2-d tensor, its row count)."""
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(0) == 0:
        t = t[0]
    t2 = t.reshape(-1, t.shape[-1])
    t2 = t2 if (t2.is_contiguous() and t2.data_ptr() % 16 == 0) else t2.contiguous()
    return t2, t2.shape[0]


def run(decoder, hidden_states, position_embeddings, reference_input, values, value_bias, keep_mask, spatial_shapes,
        level_start_index, first_with_pos=None, valid_ratios=None, keep_bits=None):
    """All layers of ``decoder``.  hidden_states / position_embeddings [B, N, 256] (stride-0 batch expansions are read in
    place), reference_input [B, N, 4, 2] (reference points x valid ratios; with ``valid_ratios`` [B, 4, 2]: the plain points [B, N, 2],
    multiplied in the kernel), values [Ld, B, S, 256] bias-free value
    projections, value_bias [Ld, 256].  Returns (states [Ld, B, N, 256], q [Ld, B, N, 256] scaled, k [Ld, B, N, 256];
    q[0] / k[0] may be stride-0 expansions over the batch).  ``keep_bits``: the bit-packed copy of ``keep_mask``
    (``ops.level_geometry``'s fifth result), handed down explicitly; packed here when absent."""
    from . import ops
    lib = _lib.lib()
    dev = hidden_states.device
    B, N, _ = hidden_states.shape
    nl = len(decoder.layers)
    S = values.shape[2]
    x0, x_rows = _rows(hidden_states, N)
    pos, pos_rows = _rows(position_embeddings, N)
    lay0 = decoder.layers[0].self_attn
    scale = float(lay0.scaling)
    # layer 0's projections: inputs are rows of the query table when both operands are batch expansions -> constants
    consts = [_layer_constants(l) for l in decoder.layers]
    with_pos0 = None
    if first_with_pos is not None:
        with_pos0, _ = _rows(first_with_pos, N)
        if with_pos0.shape[0] != x_rows:
            with_pos0 = None

    def qkv0():
        xp = with_pos0 if with_pos0 is not None else (x0 + (pos if pos_rows == x_rows else pos.repeat(x_rows // pos_rows, 1)))
        q, k, v = ops.linear_grouped([
            dict(x=xp, w=lay0.q_proj.weight, b=lay0.q_proj.bias, alpha=lay0.scaling),
            dict(x=xp, w=lay0.k_proj.weight, b=lay0.k_proj.bias),
            dict(x=x0, w=lay0.v_proj.weight, b=lay0.v_proj.bias)])
        return q.contiguous(), k.contiguous(), v.contiguous()

    def base(t):
        return t._base if t._base is not None else t

    def is_weight(t):   # a view of a module parameter (the learned query table), not a per-call activation
        return isinstance(base(t), torch.nn.Parameter)

    # ``first_with_pos`` is handed in by DeformableDetrModel.forward exactly when its inputs are DERIVED CONSTANTS of the query
    # table (ops.cached_weights "query_tables": tensors that live as long as the weights do)
    constant_inputs = first_with_pos is not None or (is_weight(hidden_states) and is_weight(position_embeddings))
    if x_rows == N and pos_rows == N and constant_inputs:
        # both operands are rows of the query table (batch expansions): the projections are derived constants, keyed on the
        # tensors the views were cut from and on the views' geometry.  Fresh ``inputs_embeds`` at B == 1 also have N rows but
        # are neither parameters nor the model's cached tables: caching those would pin one never-hit entry per call
        #  -- they take qkv0() directly
        srcs = [lay0.q_proj.weight, lay0.q_proj.bias, lay0.k_proj.weight, lay0.k_proj.bias, lay0.v_proj.weight,
                lay0.v_proj.bias, base(hidden_states), base(position_embeddings)]
        name = f"decoder_cluster_qkv0:{x0.data_ptr()}:{pos.data_ptr()}:{with_pos0 is not None}"
        q0, k0, v0 = ops.cached_weights(decoder, name, srcs, qkv0)
    else:
        q0, k0, v0 = qkv0()
    qkv_rows0 = q0.shape[0]

    states = torch.empty(nl, B * N, 256, dtype=torch.float32, device=dev)
    qkv = torch.empty(max(nl - 1, 1), 3, B * N, 256, dtype=torch.float32, device=dev)
    barriers, status, partials, ids = _workspace(dev, (B, N, nl))
    if valid_ratios is not None:   # plain [B, N, 2] points (possibly one image's rows expanded) + [B, 4, 2] ratios
        ref, ref_rows = _rows(reference_input, N)
        vr = valid_ratios.contiguous()
    else:
        ref, ref_rows, vr = reference_input.contiguous(), 0, None
    kbits = None
    if keep_mask is not None:
        from .load_custom import _check_keep_bits, pack_keep_bits
        kbits = _check_keep_bits(keep_bits, B, S) if keep_bits is not None else pack_keep_bits(keep_mask, B, S)
    vals = values if values.is_contiguous() else values.contiguous()
    vb = value_bias.contiguous() if value_bias is not None else None
    stream = _stream()
    nclusters = B * ((N + 7) // 8)
    for i, layer in enumerate(decoder.layers):
        c = consts[i]
        a = EgtrDecoderLayer()
        if i == 0:
            a.x_in, a.x_rows = x0.data_ptr(), x_rows
            a.q, a.k, a.v, a.qkv_rows = q0.data_ptr(), k0.data_ptr(), v0.data_ptr(), qkv_rows0
        else:
            a.x_in, a.x_rows = states[i - 1].data_ptr(), B * N
            a.q, a.k, a.v, a.qkv_rows = (qkv[i - 1, 0].data_ptr(), qkv[i - 1, 1].data_ptr(), qkv[i - 1, 2].data_ptr(),
                                         B * N)
        a.pos, a.pos_rows = pos.data_ptr(), pos_rows
        a.reference_points = ref.data_ptr()
        a.valid_ratios, a.ref_rows = (vr.data_ptr() if vr is not None else None), ref_rows
        a.value = vals[i].data_ptr()
        a.value_bias = vb[i].data_ptr() if vb is not None else None
        a.keep_bits = kbits.data_ptr() if kbits is not None else None
        a.spatial_shapes, a.level_start_index = spatial_shapes.data_ptr(), level_start_index.data_ptr()
        a.x_out = states[i].data_ptr()
        if i + 1 < nl:
            n = consts[i + 1]
            a.q_next, a.k_next, a.v_next = qkv[i, 0].data_ptr(), qkv[i, 1].data_ptr(), qkv[i, 2].data_ptr()
            a.w_qkv_next, a.b_qkv_next = n["w_qkv"].data_ptr(), n["b_qkv"].data_ptr()
        sa, ca = layer.self_attn, layer.encoder_attn
        a.w_attn_out, a.b_attn_out = c["w_attn_out"].data_ptr(), sa.out_proj.bias.data_ptr()
        a.ln1_gamma, a.ln1_beta = layer.self_attn_layer_norm.weight.data_ptr(), layer.self_attn_layer_norm.bias.data_ptr()
        a.w_off_logit, a.b_off_logit = c["w_off_logit"].data_ptr(), c["b_off_logit"].data_ptr()
        a.w_cross_out, a.b_cross_out = c["w_cross_out"].data_ptr(), ca.output_proj.bias.data_ptr()
        a.ln2_gamma, a.ln2_beta = (layer.encoder_attn_layer_norm.weight.data_ptr(),
                                   layer.encoder_attn_layer_norm.bias.data_ptr())
        a.w_fc1, a.b_fc1 = c["w_fc1"].data_ptr(), layer.fc1.bias.data_ptr()
        a.w_fc2, a.b_fc2 = c["w_fc2"].data_ptr(), layer.fc2.bias.data_ptr()
        a.ln3_gamma, a.ln3_beta = layer.final_layer_norm.weight.data_ptr(), layer.final_layer_norm.bias.data_ptr()
        a.partials, a.barriers, a.status, a.xcc_ids = (partials.data_ptr(), barriers.data_ptr(), status.data_ptr(),
                                                       ids.data_ptr())
        a.q_scale, a.ln_eps = scale, float(layer.self_attn_layer_norm.eps)
        a.batch, a.num_query, a.spatial_size, a.num_clusters = B, N, S, nclusters
        a.generation = 0
        _lib.check(lib.egtr_decoder_layer_f32(stream, ctypes.byref(a)), "egtr_decoder_layer_f32")
    if not torch.cuda.is_current_stream_capturing():
        if dev.index not in _CHECKED:
            st = int(status.item())   # one synchronisation, on the first eager run per device
            if st != 0:
                _reset(dev)
                raise DecoderClusterError(_describe(st) + "the per-operation decoder is used instead")
            _CHECKED.add(dev.index)
        else:
            # later runs: the sticky status words (this workspace's and every captured graph's) are polled without a
            # synchronisation every POLL_EVERY eager forwards; a time-out raises DecoderClusterError one poll late (the
            # kernel has NaN-poisoned the states it handed out in the meantime, so nothing plausible-looking leaks)
            _EAGER_CALLS[dev.index] = _EAGER_CALLS.get(dev.index, 0) + 1
            if _EAGER_CALLS[dev.index] % POLL_EVERY == 0:
                poll_status(dev)
    states = states.view(nl, B, N, 256)
    if qkv_rows0 == N:   # layer 0's projections are the same rows for every image
        q_all, k_all = [q0.unsqueeze(0).expand(B, N, 256)], [k0.unsqueeze(0).expand(B, N, 256)]
    else:
        q_all, k_all = [q0.view(B, N, 256)], [k0.view(B, N, 256)]
    for i in range(nl - 1):
        q_all.append(qkv[i, 0].view(B, N, 256))
        k_all.append(qkv[i, 1].view(B, N, 256))
    return states, q_all, k_all


def _workspaces_on(dev):
    """Every buffer set on ``dev``: eager workspaces AND the sets baked into captured graphs."""
    return ([ws for (d, _, _), ws in _WORKSPACES.items() if d == dev.index]
            + [ws for ws in _GRAPH_WORKSPACES if ws[1].device == dev])


def _status_words(dev):
    return [ws[1] for ws in _workspaces_on(dev)]


def _reset(dev):
    """After a reported failure: status, barrier counters and partial sums back to zero (stream-ordered), so that the next
    launch starts from a whole state (a barrier that lost an arrival would otherwise release early for ever after)."""
    for barriers, status, partials, _ in _workspaces_on(dev):
        status.zero_()
        barriers.zero_()
        partials.zero_()


def _describe(st):
    return ("egtr_decoder_layer_f32: " + ("a cluster barrier timed out (the decoder states of that launch are NaN-poisoned); "
                                          if st & 1 else "")
            + ("the workgroups of a cluster were spread over several XCDs; " if st & 2 else ""))


def read_status(dev):
    """The sticky status words of this device's workspaces -- eager ones and those of captured graphs -- OR-ed together
    (one synchronising copy each): 0 = every launch so far was sound."""
    st = 0
    for w in _status_words(dev):
        st |= int(w.item())
    return st


_PENDING_STATUS = {}   # device index -> (pinned int32 host word, event) of the last asynchronous poll


def poll_status(dev, wait=False):
    """Check the status words WITHOUT stalling the stream: the verdict of the PREVIOUS poll is read if its copy
    has completed (``wait=True``: is waited for), then a new asynchronous copy of the OR of all status words -- eager and
    captured-graph workspaces -- is queued on the current stream.  A non-zero word raises ``DecoderClusterError`` (bit 0: a
    cluster barrier timed out, e.g. because another stream or process held the CUs longer than the spin limit -- the kernel
    has NaN-poisoned the decoder states it handed out; bit 1: a cluster was spread over XCDs).  Callers: ``GraphedForward``
    every ``status_every`` replays, ``runtime.calculate_fps`` at the end of a run, ``runtime.triplet_candidates`` never
    (it does not synchronise either)."""
    if not isinstance(dev, torch.device):
        dev = torch.device(dev)
    if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return 0
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    verdict = 0
    pend = _PENDING_STATUS.get(dev.index)
    if pend is not None:
        host, event = pend
        if wait:
            event.synchronize()
        if event.query():
            verdict = int(host.item())
            _PENDING_STATUS.pop(dev.index, None)
    words = _status_words(dev)
    if words and dev.index not in _PENDING_STATUS:
        if len(words) == 1:
            acc = words[0]
        else:   # OR of the two-bit masks, bit by bit (a plain max would lose bit 0 next to a 2)
            stacked = torch.stack([w.reshape(()) for w in words])
            acc = ((stacked & 1).max() | (stacked & 2).max()).to(torch.int32)
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(acc.reshape(1), non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        _PENDING_STATUS[dev.index] = (host, event)
    if verdict:
        _reset(dev)
        _PENDING_STATUS.pop(dev.index, None)
        raise DecoderClusterError(_describe(verdict) + "results since the previous poll are void")
    return verdict
