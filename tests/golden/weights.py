"""Deterministic, platform-independent weight and input generation shared by the golden-vector generator
(tests/golden/make_golden.py, runs in the build container against the imported reference) and by the tests
(run anywhere).  Uses numpy's PCG64 so the committed fixtures never have to carry the multi-MB state dicts:
a fixture stores {seed, key -> shape} and the expected outputs; both sides regenerate identical weights.

Contains no reference code.
"""
import math

import numpy as np
import torch


def _scale_for(name, shape):
    """Scale rule: keeps activations O(1) through the network so parity tests are numerically meaningful."""
    if name.endswith("sampling_offsets.bias"):
        return ("normal", 2.0)  # +-2 px spread of the sampling points (plays the role of dd:999-1013)
    if name.endswith("sampling_offsets.weight"):
        return ("normal", 0.3 / math.sqrt(shape[-1]))
    if "layer_norm.weight" in name or (".input_proj." in name and name.endswith(".1.weight")) \
            or name.endswith("_norm.weight"):   # (+ the two-stage variant's enc_output_norm / pos_trans_norm)
        return ("affine", 0.1)  # 1 + 0.1 r
    if name.endswith("level_embed"):
        return ("normal", 0.5)
    if name.endswith("query_position_embeddings.weight"):
        return ("normal", 1.0)
    if name.endswith("running_var"):
        return ("positive", 0.2)  # 1 + 0.2 |r|
    if name.endswith("running_mean"):
        return ("normal", 0.1)
    if name.endswith(".bias"):
        return ("normal", 0.1)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return ("normal", 1.0 / math.sqrt(fan_in))
    return ("normal", 0.1)


_FILL_CACHE = {}


def fill_state_dict(shapes, seed, dtype=torch.float32, skip=("triplet_dist", "rel_dist"), alias_heads=True):
    """shapes: {key: tuple}. Keys are filled in sorted order from one PCG64 stream.  ``alias_heads=False``: a
    with_box_refine=True model, whose per-level heads are separate clones (egtr:148-154).  The last few results are
    memoised (several tests build the same seeded model; callers only copy the tensors into a module)."""
    key = (tuple(sorted((k, tuple(int(s) for s in v)) for k, v in shapes.items())), int(seed), str(dtype), tuple(skip),
           bool(alias_heads))
    hit = _FILL_CACHE.get(key)
    if hit is not None:
        return dict(hit)
    sd = _fill_state_dict(shapes, seed, dtype, skip, alias_heads)
    if len(_FILL_CACHE) >= 3:
        _FILL_CACHE.pop(next(iter(_FILL_CACHE)))
    _FILL_CACHE[key] = sd
    return dict(sd)


def _fill_state_dict(shapes, seed, dtype, skip, alias_heads):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k in sorted(shapes):
        if k in skip:
            continue
        shp = tuple(int(s) for s in shapes[k])
        kind, s = _scale_for(k, shp)
        r = rng.standard_normal(shp).astype(np.float64)
        if kind == "affine":
            a = 1.0 + s * r
        elif kind == "positive":
            a = 1.0 + s * np.abs(r)
        else:
            a = s * r
        sd[k] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    # class_embed.{i} / bbox_embed.{i} alias ONE module when with_box_refine=False (egtr:152-158): a real
    # checkpoint carries identical tensors under every index, so make the synthetic one do the same.
    for k in list(sd):
        for head in ("class_embed.", "bbox_embed."):
            if alias_heads and k.startswith(head):
                rest = k[len(head):].split(".", 1)[1]
                sd[k] = sd[head + "0." + rest]
        # with_box_refine=True registers the box heads a second time under the decoder (egtr:154): same tensors
        if k.startswith("model.decoder.bbox_embed."):
            sd[k] = sd[k[len("model.decoder."):]]
    return sd


def fg_matrix(num_labels, num_rel, seed=0):
    """Synthetic foreground statistics: counts in [0, 5)."""
    return np.random.RandomState(seed).randint(0, 5, (num_labels + 1, num_labels + 1, num_rel)).astype(np.float64)


def freq_bias_tables(fg, eps=1e-12):
    """triplet_dist / rel_dist exactly as the reference builds them INCLUDING its operator-precedence quirk
    (model/egtr.py:169-183: ``fg + eps / (fg.sum(2) + eps)`` then ``.log()``; use_log_softmax=False)."""
    rel_dist = torch.FloatTensor(fg.sum(axis=(0, 1)) / (fg.sum() + eps))
    triplet = torch.FloatTensor(fg + eps / (fg.sum(2, keepdims=True) + eps)).log()
    return triplet, rel_dist


def rng_inputs(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_msda_inputs(seed, B, Lq, M, D, shapes, P, dtype=torch.float32, oob_frac=0.08):
    """Random MSDA operands: loc spread around [0,1] with a fraction pushed outside (exercises cuh:288 and the
    per-corner range checks), softmax-normalised attention weights."""
    rng = rng_inputs(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rng.standard_normal((B, S, M, D))
    loc = rng.uniform(-0.05, 1.05, (B, Lq, M, L, P, 2))
    far = rng.uniform(0, 1, loc.shape[:-1]) < oob_frac
    loc[far] += rng.choice([-1.5, 1.5], size=(int(far.sum()), 2))
    a = rng.standard_normal((B, Lq, M, L * P))
    a = np.exp(a - a.max(-1, keepdims=True))
    a = (a / a.sum(-1, keepdims=True)).reshape(B, Lq, M, L, P)
    grad_out = rng.standard_normal((B, Lq, M * D))
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dtype)  # noqa: E731
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    return dict(value=t(value), shapes=shp, lsi=lsi, loc=t(loc), attn=t(a), grad_out=t(grad_out))


def make_targets(seed, B, N, C, R, tmin=3, tmax=8):
    """Synthetic SGG labels: T boxes per image, 3T random relations in the dense [N,N,R] target."""
    rng = rng_inputs(seed)
    targets = []
    for _ in range(B):
        T = int(rng.integers(tmin, tmax + 1))
        cxcy = rng.uniform(0.2, 0.8, (T, 2))
        wh = rng.uniform(0.05, 0.25, (T, 2))
        labels = rng.integers(0, C, (T,))
        rel = np.zeros((N, N, R), dtype=np.float32)
        for _r in range(3 * T):
            s, o = rng.integers(0, T, (2,))
            if s == o:
                continue
            rel[s, o, int(rng.integers(0, R))] = 1.0
        targets.append(dict(class_labels=torch.from_numpy(labels.astype(np.int64)),
                            boxes=torch.from_numpy(np.concatenate([cxcy, wh], 1).astype(np.float32)),
                            rel=torch.from_numpy(rel)))
    return targets


def edge_targets(kind, N, C, R):
    """Target sets at the edges of what the criterion sees (used with the model / inputs of sgg_small.npz):
    "empty" = image 1 without any annotated object, "single" = image 0 with ONE object and no relation,
    "crowded" = image 0 with 20 objects (of N = 24 queries) and a dense relation tensor."""
    targets = make_targets(23, 2, N, C, R)
    rng = rng_inputs(77)
    if kind == "empty":
        t = targets[1]
        t["class_labels"], t["boxes"], t["rel"] = t["class_labels"][:0], t["boxes"][:0], torch.zeros(N, N, R)
    elif kind == "single":
        t = targets[0]
        t["class_labels"], t["boxes"], t["rel"] = t["class_labels"][:1], t["boxes"][:1], torch.zeros(N, N, R)
    elif kind == "crowded":
        n = 20
        cxcy = rng.uniform(0.2, 0.8, (n, 2))
        wh = rng.uniform(0.05, 0.3, (n, 2))
        rel = np.zeros((N, N, R), dtype=np.float32)
        rel[:n, :n] = (rng.uniform(0, 1, (n, n, R)) < 0.15).astype(np.float32)
        rel[np.arange(n), np.arange(n)] = 0
        targets[0] = dict(class_labels=torch.from_numpy(rng.integers(0, C, (n,)).astype(np.int64)),
                          boxes=torch.from_numpy(np.concatenate([cxcy, wh], 1).astype(np.float32)),
                          rel=torch.from_numpy(rel))
    else:
        raise ValueError(kind)
    return targets


def post_inputs(seed=61, B=2, N=200, C=150, R=50):
    """Seeded model outputs + targets for the post-processing fixtures (evaluate_batch, train_egtr.py:43-106).
    Image 1's relation scores are quantised to multiples of 1/8 and its connectivity to {0.5, 1}, so exactly tied
    triplet scores occur (argsort order inside a tie group is implementation-defined)."""
    rng = rng_inputs(seed)
    logits = torch.from_numpy(rng.standard_normal((B, N, C + 1)) * 2).float()
    cxcy = rng.uniform(0.15, 0.85, (B, N, 2))
    wh = rng.uniform(0.04, 0.3, (B, N, 2))
    boxes = torch.from_numpy(np.concatenate([cxcy, wh], -1)).float()
    rel = torch.from_numpy(rng.uniform(0, 1, (B, N, N, R))).float()
    conn = torch.from_numpy(rng.uniform(0, 1, (B, N, N, 1))).float()
    rel[1] = torch.round(rel[1] * 8) / 8
    conn[1] = torch.where(conn[1] > 0.5, torch.ones(()), torch.full((), 0.5))
    outputs = {"logits": logits, "pred_boxes": boxes, "pred_rel": rel, "pred_connectivity": conn}
    sizes = [(600, 1000), (480, 640)]
    targets = []
    for b, t in enumerate(make_targets(seed + 100, B, N, C, R, tmin=5, tmax=30)):
        t = dict(t)
        t["orig_size"] = torch.tensor(sizes[b])
        targets.append(t)
    return outputs, targets, dict(num_labels=C, sizes=sizes)


def bbox_cases(seed=62):
    """xyxy box sets for bbox_overlaps / bbox_intersections (bbox.pyx:21-108): random, integer-aligned with touching /
    one-pixel-overlap pairs (the +1 convention makes touching boxes overlap), degenerate and empty inputs."""
    rng = rng_inputs(seed)

    def rnd(n, size):
        x0 = rng.uniform(0, size * 0.8, (n, 1))
        y0 = rng.uniform(0, size * 0.8, (n, 1))
        return np.concatenate([x0, y0, x0 + rng.uniform(1, size * 0.4, (n, 1)), y0 + rng.uniform(1, size * 0.4, (n, 1))],
                              1)

    grid = np.array([[0, 0, 9, 9], [10, 0, 19, 9], [9, 9, 20, 20], [0, 10, 9, 19], [5, 5, 5, 5], [20, 20, 30, 30],
                     [-5, -5, 4, 4], [3, 3, 2, 2]], dtype=np.float64)
    return {"random": (rnd(200, 1000.0), rnd(37, 1000.0)), "grid": (grid, grid.copy()),
            "ints": (np.floor(rnd(64, 64.0)), np.floor(rnd(64, 64.0))),
            "empty_q": (rnd(5, 100.0), np.zeros((0, 4)))}
