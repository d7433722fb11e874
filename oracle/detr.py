"""Oracle: Deformable-DETR encoder/decoder + EGTR relation head, functional CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Everything here is a pure function of
``(state_dict, config, inputs)``; the state-dict keys are the reference's own, so a
reference checkpoint, the product model and this oracle all consume the same flat dict.

Citations are to /root/reference (model/deformable_detr.py = "dd", model/egtr.py = "egtr").
Inference semantics (dropout inactive) unless ``cfg["dropout"] == 0`` makes train/eval identical anyway.
"""
import math

import torch
import torch.nn.functional as F

from .msda import msda_forward_grid_sample


# --------------------------------------------------------------------------------------- small helpers
def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mlp(sd, p, x, n):
    """DeformableDetrMLPPredictionHead (dd:2865-2883): ReLU between layers, none after the last."""
    for i in range(n):
        x = _lin(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def inverse_sigmoid(x, eps=1e-5):
    """dd:658-662."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def sine_position_embedding(mask, embedding_dim=128, temperature=10000.0, scale=2 * math.pi):
    """DeformableDetrSinePositionEmbedding with normalize=True (dd:850-876). mask [B,H,W] bool -> [B,2E,H,W]."""
    y_embed = mask.cumsum(1, dtype=torch.float32)
    x_embed = mask.cumsum(2, dtype=torch.float32)
    eps = 1e-6
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(embedding_dim, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / embedding_dim)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------- attention modules
def msda_module(sd, p, hidden, enc_states, mask, pos, refs, shapes, lsi, core=None,
                n_heads=8, n_levels=4, n_points=4, return_parts=False):
    """DeformableDetrMultiscaleDeformableAttention.forward (dd:1026-1104)."""
    if pos is not None:
        hidden = hidden + pos  # :1039-1040
    B, Lq, d = hidden.shape
    S = enc_states.shape[1]
    value = _lin(sd, p + ".value_proj", enc_states)  # :1049
    if mask is not None:
        value = value.masked_fill(~mask[..., None], 0.0)  # :1052
    value = value.view(B, S, n_heads, d // n_heads)
    off = _lin(sd, p + ".sampling_offsets", hidden).view(B, Lq, n_heads, n_levels, n_points, 2)
    aw = _lin(sd, p + ".attention_weights", hidden).view(B, Lq, n_heads, n_levels * n_points)
    aw = F.softmax(aw, -1).view(B, Lq, n_heads, n_levels, n_points)  # :1062
    if refs.shape[-1] == 2:
        norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)  # (W, H) :1067-1069
        loc = refs[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = refs[:, :, None, :, None, :2] + off / n_points * refs[:, :, None, :, None, 2:] * 0.5
    if core is None:
        out = msda_forward_grid_sample(value, shapes, loc, aw)  # fallback path :1099
    else:
        out = core(value, shapes, lsi, loc, aw)
    res = _lin(sd, p + ".output_proj", out)
    if return_parts:
        return res, dict(value=value, loc=loc, attn=aw, core_out=out)
    return res


def decoder_self_attention(sd, p, x, pos, n_heads=8):
    """DeformableDetrMultiheadAttention.forward (dd:1149-1262), no mask, dropout inactive.

    Returns (out [B,N,d], q_scaled [B,M,N,D], k [B,M,N,D])."""
    B, N, d = x.shape
    D = d // n_heads
    xp = x + pos
    q = _lin(sd, p + ".q_proj", xp) * (D ** -0.5)  # :1166
    k = _lin(sd, p + ".k_proj", xp)
    v = _lin(sd, p + ".v_proj", x)  # hidden_states_original :1168

    def shape(t):
        return t.view(B, N, n_heads, D).transpose(1, 2).contiguous()

    qh, kh, vh = shape(q), shape(k), shape(v)
    w = torch.matmul(qh, kh.transpose(-1, -2))  # :1190
    w = F.softmax(w, dim=-1)  # :1217
    o = torch.matmul(w, vh)  # :1237
    o = o.transpose(1, 2).reshape(B, N, d)
    return _lin(sd, p + ".out_proj", o), qh, kh


def encoder_layer(sd, p, x, mask, pos, refs, shapes, lsi, core=None):
    """DeformableDetrEncoderLayer.forward (dd:1283-1358), eval mode."""
    a = msda_module(sd, p + ".self_attn", x, x, mask, pos, refs, shapes, lsi, core)
    x = _ln(sd, p + ".self_attn_layer_norm", x + a)
    h = F.relu(_lin(sd, p + ".fc1", x))
    h = _lin(sd, p + ".fc2", h)
    return _ln(sd, p + ".final_layer_norm", x + h)


def decoder_layer(sd, p, x, pos, refs, shapes, lsi, enc, enc_mask, core=None):
    """DeformableDetrDecoderLayer.forward (dd:1390-1489), eval mode. Returns (x, q_scaled, k)."""
    a, q, k = decoder_self_attention(sd, p + ".self_attn", x, pos)
    x = _ln(sd, p + ".self_attn_layer_norm", x + a)
    c = msda_module(sd, p + ".encoder_attn", x, enc, enc_mask, pos, refs, shapes, lsi, core)
    x = _ln(sd, p + ".encoder_attn_layer_norm", x + c)
    h = F.relu(_lin(sd, p + ".fc1", x))
    h = _lin(sd, p + ".fc2", h)
    return _ln(sd, p + ".final_layer_norm", x + h), q, k


def encoder_reference_points(shapes, valid_ratios):
    """DeformableDetrEncoder.get_reference_points (dd:1616-1648)."""
    pts = []
    for lvl, (H, W) in enumerate(shapes.tolist()):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32),
                                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        pts.append(torch.stack((rx, ry), -1))
    pts = torch.cat(pts, 1)
    return pts[:, :, None] * valid_ratios[:, None]


def valid_ratio(mask):
    """DeformableDetrModel.get_valid_ratio (dd:2064-2073). mask [B,H,W] bool -> [B,2] (w, h)."""
    _, H, W = mask.shape
    vh = torch.sum(mask[:, :, 0], 1).float() / H
    vw = torch.sum(mask[:, 0, :], 1).float() / W
    return torch.stack([vw, vh], -1)


# --------------------------------------------------------------------------------------- backbones
def stub_backbone(sd, pixel_values, pixel_mask, prefix="model.backbone.conv_encoder.model"):
    """The golden fixtures' stand-in backbone (tests/golden/_ref_import.py StubBackbone): three 1x1 convs on
    strided views, strides 8/16/32."""
    out = []
    for i, s in enumerate((8, 16, 32)):
        f = F.conv2d(pixel_values[:, :, ::s, ::s], sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"])
        m = F.interpolate(pixel_mask[None].float(), size=f.shape[-2:]).to(torch.bool)[0]
        out.append((f, m))
    return out


def _frozen_bn(sd, p, x):
    """DeformableDetrFrozenBatchNorm2d.forward (dd:704-714)."""
    w = sd[p + ".weight"].reshape(1, -1, 1, 1)
    b = sd[p + ".bias"].reshape(1, -1, 1, 1)
    rv = sd[p + ".running_var"].reshape(1, -1, 1, 1)
    rm = sd[p + ".running_mean"].reshape(1, -1, 1, 1)
    scale = w * (rv + 1e-5).rsqrt()
    return x * scale + (b - rm * scale)


def resnet50_backbone(sd, pixel_values, pixel_mask, prefix="model.backbone.conv_encoder.model"):
    """ResNet-50 (timm ``features_only`` naming, out_indices 2,3,4 = layer2..layer4; dd:748-787) with frozen BN.
    timm itself is absent from the container, so parity of this block with timm is unpinned."""
    def bn(p, x):
        return _frozen_bn(sd, f"{prefix}.{p}", x)

    x = F.conv2d(pixel_values, sd[f"{prefix}.conv1.weight"], None, stride=2, padding=3)
    x = F.relu(bn("bn1", x))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = []
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for b in range(blocks):
            p = f"layer{li}.{b}"
            s = stride if b == 0 else 1
            idt = x
            y = F.relu(bn(p + ".bn1", F.conv2d(x, sd[f"{prefix}.{p}.conv1.weight"])))
            y = F.relu(bn(p + ".bn2", F.conv2d(y, sd[f"{prefix}.{p}.conv2.weight"], stride=s, padding=1)))
            y = bn(p + ".bn3", F.conv2d(y, sd[f"{prefix}.{p}.conv3.weight"]))
            if f"{prefix}.{p}.downsample.0.weight" in sd:
                idt = bn(p + ".downsample.1", F.conv2d(x, sd[f"{prefix}.{p}.downsample.0.weight"], stride=s))
            x = F.relu(y + idt)
        if li >= 2:
            m = F.interpolate(pixel_mask[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
            feats.append((x, m))
    return feats


# --------------------------------------------------------------------------------------- full model
def proposal_pos_embed(proposals):
    """dd:2075-2096: sine embedding of proposal logits [B, K, 4] -> [B, K, 512]."""
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / 128)
    pos = (proposals.sigmoid() * (2 * math.pi))[:, :, :, None] / dim_t
    return torch.stack((pos[:, :, :, 0::2].sin(), pos[:, :, :, 1::2].cos()), dim=4).flatten(2)


def encoder_output_proposals(sd, enc, padding_mask, shapes):
    """dd:2098-2159: per-token proposals (pixel centre over the VALID extent, size 0.05 * 2^level) as logits, +inf where the
    token is padded or the proposal leaves (0.01, 0.99); features of those tokens zeroed, then enc_output + LayerNorm."""
    B = enc.shape[0]
    props, cur = [], 0
    for lvl, (H, W) in enumerate(shapes.tolist()):
        m = padding_mask[:, cur:cur + H * W].view(B, H, W)
        vh, vw = (~m[:, :, 0]).sum(1), (~m[:, 0, :]).sum(1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        grid = torch.stack([gx, gy], -1)[None].expand(B, -1, -1, -1)
        grid = (grid + 0.5) / torch.stack([vw, vh], 1).view(B, 1, 1, 2)
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat([grid, wh], -1).view(B, -1, 4))
        cur += H * W
    props = torch.cat(props, 1)
    valid = ((props > 0.01) & (props < 0.99)).all(-1, keepdim=True)
    props = torch.log(props / (1 - props))
    drop = padding_mask.unsqueeze(-1) | ~valid
    props = props.masked_fill(drop, float("inf"))
    feat = enc.masked_fill(drop, 0.0)
    return _ln(sd, "model.enc_output_norm", _lin(sd, "model.enc_output", feat)), props


def detr_model_forward(sd, cfg, pixel_values, pixel_mask, backbone=stub_backbone, core=None):
    """DeformableDetrModel.forward (dd:2161-2390); ``cfg["with_box_refine"]`` turns the decoder's iterative box
    refinement on (dd:1903-1918), ``cfg["two_stage"]`` the per-token proposal branch (dd:2306-2337).

    Returns dict with encoder_last_hidden_state, intermediate_hidden_states [B,Ld,N,d],
    init_reference_points [B,N,2], intermediate_reference_points [B,Ld,N,2], queries/keys tuples."""
    d = cfg["d_model"]
    L = cfg["num_feature_levels"]
    B = pixel_values.shape[0]
    feats = backbone(sd, pixel_values, pixel_mask)
    pos_list = [sine_position_embedding(m, d // 2) for _, m in feats]
    srcs, masks = [], []
    for lvl, (f, m) in enumerate(feats):  # :2221-2225
        s = F.conv2d(f, sd[f"model.input_proj.{lvl}.0.weight"], sd[f"model.input_proj.{lvl}.0.bias"])
        s = F.group_norm(s, 32, sd[f"model.input_proj.{lvl}.1.weight"], sd[f"model.input_proj.{lvl}.1.bias"], 1e-5)
        srcs.append(s)
        masks.append(m)
    for lvl in range(len(feats), L):  # :2228-2241
        inp = feats[-1][0] if lvl == len(feats) else srcs[-1]
        s = F.conv2d(inp, sd[f"model.input_proj.{lvl}.0.weight"], sd[f"model.input_proj.{lvl}.0.bias"],
                     stride=2, padding=1)
        s = F.group_norm(s, 32, sd[f"model.input_proj.{lvl}.1.weight"], sd[f"model.input_proj.{lvl}.1.bias"], 1e-5)
        m = F.interpolate(pixel_mask[None].float(), size=s.shape[-2:]).to(torch.bool)[0]
        srcs.append(s)
        masks.append(m)
        pos_list.append(sine_position_embedding(m, d // 2).to(s.dtype))
    src_f, mask_f, pos_f, shapes = [], [], [], []
    for lvl, (s, m, pe) in enumerate(zip(srcs, masks, pos_list)):  # :2253-2265
        shapes.append((s.shape[2], s.shape[3]))
        src_f.append(s.flatten(2).transpose(1, 2))
        mask_f.append(m.flatten(1))
        pos_f.append(pe.flatten(2).transpose(1, 2) + sd["model.level_embed"][lvl].view(1, 1, -1))
    src = torch.cat(src_f, 1)
    mask = torch.cat(mask_f, 1)
    pos = torch.cat(pos_f, 1)
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    vr = torch.stack([valid_ratio(m) for m in masks], 1).float()  # :2275-2278

    refs = encoder_reference_points(shapes, vr)
    x = src
    for i in range(cfg["encoder_layers"]):
        x = encoder_layer(sd, f"model.encoder.layers.{i}", x, mask, pos, refs, shapes, lsi, core)
    enc = x

    enc_cls = enc_box = None
    if cfg.get("two_stage", False):  # :2306-2337
        Ld = cfg["decoder_layers"]
        feat, props = encoder_output_proposals(sd, enc, ~mask, shapes)
        enc_cls = _lin(sd, f"class_embed.{Ld}", feat)
        enc_box = _mlp(sd, f"bbox_embed.{Ld}", feat, 3) + props
        top = torch.topk(enc_cls[..., 0], cfg["two_stage_num_proposals"], dim=1)[1]
        top_logits = torch.gather(enc_box, 1, top.unsqueeze(-1).repeat(1, 1, 4)).detach()
        ref = top_logits.sigmoid()
        pt = _ln(sd, "model.pos_trans_norm", _lin(sd, "model.pos_trans", proposal_pos_embed(top_logits)))
        query_embed, target = torch.split(pt, d, dim=2)
    else:
        qe = sd["model.query_position_embeddings.weight"]
        query_embed, target = torch.split(qe, d, dim=1)  # :2339
        query_embed = query_embed.unsqueeze(0).expand(B, -1, -1)
        target = target.unsqueeze(0).expand(B, -1, -1)
        ref = _lin(sd, "model.reference_points", query_embed).sigmoid()  # :2342
    h = target
    inter, inter_ref, qs, ks = [], [], [], []
    init_ref = ref
    for i in range(cfg["decoder_layers"]):
        if ref.shape[-1] == 4:  # :1857-1862 (after the first refinement step)
            ref_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
        else:
            ref_in = ref[:, :, None] * vr[:, None]  # :1865-1867
        h, q, k = decoder_layer(sd, f"model.decoder.layers.{i}", h, query_embed, ref_in, shapes, lsi, enc, mask, core)
        if cfg.get("with_box_refine", False):  # iterative box refinement, :1903-1918 (egtr:152-154 hands the heads in)
            tmp = _mlp(sd, f"bbox_embed.{i}", h, 3)
            if ref.shape[-1] == 4:
                new_ref = (tmp + inverse_sigmoid(ref)).sigmoid()
            else:
                new_ref = torch.cat([tmp[..., :2] + inverse_sigmoid(ref), tmp[..., 2:]], -1).sigmoid()
            ref = new_ref.detach()
        inter.append(h)
        inter_ref.append(ref)
        qs.append(q)
        ks.append(k)
    return dict(encoder_last_hidden_state=enc, last_hidden_state=h,
                intermediate_hidden_states=torch.stack(inter, 1), init_reference_points=init_ref,
                intermediate_reference_points=torch.stack(inter_ref, 1),
                decoder_attention_queries=tuple(qs), decoder_attention_keys=tuple(ks),
                spatial_shapes=shapes, level_start_index=lsi, valid_ratios=vr, mask_flatten=mask,
                enc_outputs_class=enc_cls, enc_outputs_coord_logits=enc_box)


def detection_heads(sd, cfg, mo):
    """egtr:283-314 (class_embed / bbox_embed indices alias one module when with_box_refine=False)."""
    hs = mo["intermediate_hidden_states"]
    logits_all, boxes_all = [], []
    for lvl in range(hs.shape[1]):
        ref = mo["init_reference_points"] if lvl == 0 else mo["intermediate_reference_points"][:, lvl - 1]
        ref = inverse_sigmoid(ref)
        lg = _lin(sd, f"class_embed.{lvl}", hs[:, lvl])
        bx = _mlp(sd, f"bbox_embed.{lvl}", hs[:, lvl], 3).clone()
        if ref.shape[-1] == 4:
            bx = bx + ref  # :294-295 (refined references)
        else:
            bx[..., :2] += ref  # :297
        logits_all.append(lg)
        boxes_all.append(bx.sigmoid())
    return torch.stack(logits_all, 1), torch.stack(boxes_all, 1)


def relation_head(sd, cfg, queries, keys, last_hidden, logits):
    """EGTR relation head in the reference's evaluation order (egtr:322-418): materialises
    relation_source [B,N,N,Ld+1,2d], gate = sigmoid(Linear(2d->1)), gated sum over slots, two 3-layer MLPs.

    Returns (pred_rel_logits incl. freq bias [B,N,N,R], pred_connectivity_logits [B,N,N,1], gate [B,N,N,Ld+1,1])."""
    B, N, d = last_hidden.shape
    D = d // cfg["encoder_attention_heads"]
    unscale = D ** 0.5
    pq = [_lin(sd, f"proj_q.{l}", q.transpose(1, 2).reshape(B, N, d) * unscale) for l, q in enumerate(queries)]
    pk = [_lin(sd, f"proj_k.{l}", k.transpose(1, 2).reshape(B, N, d)) for l, k in enumerate(keys)]
    Q = torch.stack(pq, -2)  # [B,N,Ld,d]
    K = torch.stack(pk, -2)
    Qr = Q.unsqueeze(2).repeat(1, 1, N, 1, 1)
    Kr = K.unsqueeze(1).repeat(1, N, 1, 1, 1)
    src = torch.cat([Qr, Kr], dim=-1)  # :373-375
    sub = _lin(sd, "final_sub_proj", last_hidden).unsqueeze(2).repeat(1, 1, N, 1)
    obj = _lin(sd, "final_obj_proj", last_hidden).unsqueeze(1).repeat(1, N, 1, 1)
    src = torch.cat([src, torch.cat([sub, obj], dim=-1).unsqueeze(-2)], dim=-2)  # :390-396
    gate = torch.sigmoid(_lin(sd, "rel_predictor_gate", src))  # :400
    z = torch.mul(gate, src).sum(dim=-2)  # :401
    rel = _mlp(sd, "rel_predictor", z, 3)
    if cfg.get("use_freq_bias", True):  # :405-413
        node = torch.argmax(logits, dim=-1)
        rel = rel + torch.stack([sd["triplet_dist"][node[i]][:, node[i]] for i in range(B)], 0)
    conn = _mlp(sd, "connectivity_layer", z, 3)
    return rel, conn, gate


def sgg_forward(sd, cfg, pixel_values, pixel_mask, backbone=stub_backbone, core=None):
    """DetrForSceneGraphGeneration.forward without labels (egtr:241-540). Returns a dict whose
    ``pred_rel`` / ``pred_connectivity`` are post-sigmoid like the reference's output object, plus the
    pre-sigmoid ``rel_logits`` / ``conn_logits`` the loss consumes (egtr:450-454)."""
    mo = detr_model_forward(sd, cfg, pixel_values, pixel_mask, backbone, core)
    logits_all, boxes_all = detection_heads(sd, cfg, mo)
    logits, boxes = logits_all[:, -1], boxes_all[:, -1]
    rel, conn, gate = relation_head(sd, cfg, mo["decoder_attention_queries"], mo["decoder_attention_keys"],
                                    mo["last_hidden_state"], logits)
    rel_out = rel
    if cfg.get("logit_adjustment", False):  # egtr:509-512
        rel_out = rel - cfg["logit_adj_tau"] * sd["rel_dist"].log()
    out = dict(mo)
    out.update(logits=logits, pred_boxes=boxes, logits_all=logits_all, boxes_all=boxes_all,
               rel_logits=rel, conn_logits=conn, rel_gate=gate,
               pred_rel=rel_out.sigmoid(), pred_connectivity=conn.sigmoid())
    return out
