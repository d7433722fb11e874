"""Run-time helpers around the model: HIP-graph replay of the inference forward, the reference's optimizer
grouping, and one-process-per-GPU data parallelism over RCCL (``torch.distributed`` backend "nccl" on ROCm).

The reference drives training through pytorch-lightning (train_egtr.py:770-783: DDPStrategy, accumulate 2,
clip 0.1) and measures FPS with a bare loop (evaluate_egtr.py:26-36).  Lightning is not part of the hot path and
is absent on the GPU box; these helpers reproduce exactly the pieces that touch it.
"""
import collections
import contextlib
import os

import torch
import torch.distributed as dist


class _GraphEntry:
    """One captured forward: the graph, its static input buffers and static outputs, and the weight epoch it was captured at."""
    __slots__ = ("graph", "static_in", "static_out", "epoch", "replays")

    def __init__(self, graph, static_in, static_out, epoch):
        self.graph, self.static_in, self.static_out, self.epoch, self.replays = graph, static_in, static_out, epoch, 0


class GraphedForward:
    """Replay ``model(pixel_values, pixel_mask, output_attention_states=True, output_hidden_states=True)`` from captured
    HIP graphs.  At bs = 1 the forward is ~150 short kernels, i.e. launch-bound when issued eagerly
    (MI355X_MICROARCH.md: eager launch ~3.3-3.8 us host time each); a graph replay issues them back-to-back.
    Inputs are copied into static buffers; outputs are the graph's static tensors (valid until the next call).

    **Variable image sizes** (the reference's FPS loop runs over a dataloader whose images are resized to a short side of
    ``--min_size`` / long side <= ``--max_size``, evaluate_egtr.py:26-36, 165-175, so almost every batch has its own shape):
    the object keeps an LRU of up to ``max_graphs`` captured graphs keyed by (pixel shape, mask shape, dtypes, device).  A
    shape seen before replays its own graph -- own static inputs, own outputs, own private memory pool (a 600x1000 fp32
    forward holds < 0.5 GB; sixteen of them are noise in 288 GB of HBM) -- and only a NEW shape pays the two eager
    warm-up runs + capture (the warm-up runs are also what lets MIOpen / TunableOp pick their kernels for the new
    convolution / GEMM shapes outside the capture).  The least recently used graph is dropped when the table is full;
    ``capture_after = k`` defers a shape's capture until it is seen for the k-th time (eager launches before that).

    **Weights.**  A captured graph bakes in pointers to DERIVED tensors built at capture time (folded-BN backbone weights,
    stacked / concatenated projection weights, query tables).  The table is therefore tagged with a weight EPOCH: the full
    fingerprint of the model's parameters and buffers (storage pointer + version counter of each, ~140 us of host time for
    the 589 tensors of EGTR) is taken once per epoch, not per call.  What starts a new epoch: ``load_state_dict`` (a hook
    on the model), an explicit ``invalidate()``, and -- as the per-call O(1) check -- a change of the version counter or
    storage of any of a few SENTINEL tensors (the first and last parameter of the backbone, encoder, decoder and heads: an
    optimizer step, ``.to()`` / ``.half()`` / ``.cuda()`` or an in-place edit of the whole model moves all of them).  Every ``verify_every`` calls the full fingerprint is compared again as a backstop for edits that touch
    only non-sentinel tensors.  On a new epoch every graph is dropped (they all hold stale constants).  Edits through
    ``.data`` bypass the version counter -- call ``invalidate()`` after those.

    ``strict=True`` (what bench.py uses): a failed capture raises.  ``strict=False``: falls back to eager launches
    (``self.graphed = False``, reason in ``capture_error``) -- same kernels either way, about 2x slower at bs = 1.

    Every ``status_every`` calls the sticky status words of the one-launch decoder layers are polled without a
    synchronisation (``decoder_fused.poll_status``: asynchronous copy now, verdict at the next poll); a cluster barrier
    that timed out raises ``DecoderClusterError`` instead of handing out void decoder states."""

    def __init__(self, model, enabled=True, warmup=2, strict=False, max_graphs=16, verify_every=256, status_every=64,
                 capture_after=1, bucket=None):
        self.model = model
        self.enabled = enabled
        self.warmup = warmup
        self.strict = strict
        self.max_graphs = max(1, int(max_graphs))
        self.verify_every = max(1, int(verify_every))
        self.status_every = max(1, int(status_every))
        # a shape is captured when it is seen for the `capture_after`-th time and runs eagerly before that: a dataloader with
        # hundreds of distinct shapes (every long side between 600 and 1000) should not pay ~1.5 s of warm-up + capture for
        # a shape it meets once -- raise it (and `max_graphs`: sixteen graphs are ~6 GB of the 288) for such streams
        self.capture_after = max(1, int(capture_after))
        # bucket = k (opt-in): images are placed in the top-left corner of a zero canvas whose height and width are rounded up
        # to multiples of k, the rest masked out in pixel_mask -- exactly what the reference's collate does to the smaller
        # images of a BATCH (DetrFeatureExtractor.pad_and_create_pixel_mask) -- so that a dataset with hundreds of distinct
        # sizes needs a few dozen graphs (k = 32: <= 14 long sides for short side 600 / long side 600..1000, x 2
        # orientations).  Outputs are those of the padded forward (normalised boxes refer to the VALID region through the
        # valid ratios, as in batched inference), not bit-identical to the unpadded one.
        self.bucket = int(bucket) if bucket else None
        self._seen = collections.Counter()
        self.eager_calls = 0
        self._entries = collections.OrderedDict()   # key -> _GraphEntry, least recently used first
        self._tensors = None
        self._sentinels = None
        self._sentinel_key = None
        self._fingerprint_key = None
        self._epoch = 0
        self._calls = 0
        self.graphed = False
        self.captures = 0          # graphs captured so far (a shape that comes back after eviction counts again)
        self.evictions = 0
        self.capture_error = None
        self._hooks = []
        self._install_hooks()

    # ---- weight epochs -------------------------------------------------------------------------------------------
    def _install_hooks(self):
        """load_state_dict starts a new epoch at once (it also bumps every version counter, which the sentinels see);
        ``.to()`` / ``.cuda()`` / ``.half()`` replace every parameter's storage, which the sentinels see as well."""
        bump = self._bump
        try:
            self._hooks.append(self.model.register_load_state_dict_post_hook(lambda module, incompatible: bump()))
        except Exception:  # pragma: no cover - torch without the hook: the sentinels / verify pass still catch it
            pass

    def _bump(self):
        self._epoch += 1
        self._tensors = None
        self._sentinels = None

    def _all_tensors(self):
        if self._tensors is None:
            self._tensors = list(self.model.parameters()) + list(self.model.buffers())
        return self._tensors

    def _fingerprint(self):
        return [(t.data_ptr(), t._version) for t in self._all_tensors()]

    def _sentinel_fingerprint(self):
        if self._sentinels is None:
            picks = []
            for child in self.model.children():
                ps = list(child.parameters())
                if ps:
                    picks += [ps[0], ps[-1]]
                for grand in child.children():
                    gs = list(grand.parameters())
                    if gs:
                        picks += [gs[0], gs[-1]]
            own = list(self.model.parameters())
            picks += own[:1] + own[-1:]
            seen, uniq = set(), []
            for t in picks:
                if id(t) not in seen:
                    seen.add(id(t))
                    uniq.append(t)
            self._sentinels = uniq[:24]
        return tuple((t.data_ptr(), t._version) for t in self._sentinels)

    def _weights_changed(self):
        """O(1) per call; the full fingerprint only when an epoch was announced, a sentinel moved, or on the periodic verify."""
        full = self._fingerprint_key is None or self._calls % self.verify_every == 0
        epoch_now = self._epoch
        if self._sentinels is None or self._sentinel_fingerprint() != self._sentinel_key:
            full = True
        if getattr(self, "_seen_epoch", None) != epoch_now:
            full = True
        if not full:
            return False
        self._seen_epoch = epoch_now
        self._tensors = self._sentinels = None     # re-enumerate: a parameter OBJECT may have been replaced
        self._sentinel_key = self._sentinel_fingerprint()
        fp = self._fingerprint()
        if fp == self._fingerprint_key:
            return False
        changed = self._fingerprint_key is not None
        self._fingerprint_key = fp
        return changed

    def invalidate(self):
        """Forget every captured graph and every derived constant (after weights were edited through ``.data``)."""
        from . import ops
        ops.invalidate_derived(self.model)
        self._drop_all()
        self._bump()
        self._fingerprint_key = None

    def _drop_all(self):
        self._entries.clear()

    # ---- capture / replay ----------------------------------------------------------------------------------------
    def _eager(self, pv, pm):
        return self.model(pixel_values=pv, pixel_mask=pm, output_attentions=False, output_attention_states=True,
                          output_hidden_states=True)

    @staticmethod
    def _key(pv, pm):
        return (tuple(pv.shape), pv.dtype, tuple(pm.shape), pm.dtype, pv.device)

    def _capture(self, pv, pm):
        static_in = (pv.clone(), pm.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._eager(*static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_out = self._eager(*static_in)
        self.captures += 1
        self.graphed = True
        return _GraphEntry(g, static_in, static_out, self._epoch)

    @property
    def cached_shapes(self):
        """(pixel shape, ...) keys of the graphs held now, least recently used first."""
        return [k[0] for k in self._entries]

    def prime(self, shapes, dtype=torch.float32, device=None):
        """Capture a graph for every (B, 3, H, W) in ``shapes`` ahead of time (zeros as pixels, all-ones masks) -- what an
        evaluation run does once for the resize buckets of its dataloader; returns the number of new captures."""
        device = device if device is not None else next(self.model.parameters()).device
        before = self.captures
        for shp in shapes:
            b, _, h, w = shp
            self(torch.zeros(*shp, dtype=dtype, device=device), torch.ones(b, h, w, dtype=torch.long, device=device))
        return self.captures - before

    def _to_bucket(self, pv, pm):
        k = self.bucket
        h, w = pv.shape[-2:]
        hb, wb = -(-h // k) * k, -(-w // k) * k
        if (hb, wb) == (h, w):
            return pv, pm
        cpv = pv.new_zeros(*pv.shape[:-2], hb, wb)
        cpv[..., :h, :w] = pv
        cpm = pm.new_zeros(*pm.shape[:-2], hb, wb)
        cpm[..., :h, :w] = pm
        return cpv, cpm

    def _poll_decoder_status(self, device):
        from . import decoder_fused
        decoder_fused.poll_status(device)

    @torch.no_grad()
    def __call__(self, pv, pm):
        if not self.enabled:
            return self._eager(pv, pm)
        if self.bucket:
            pv, pm = self._to_bucket(pv, pm)
        self._calls += 1
        if self._weights_changed():
            self._drop_all()      # every graph holds constants derived from the old weights
        key = self._key(pv, pm)
        entry = self._entries.get(key)
        if entry is None and self.capture_after > 1:
            self._seen[key] += 1
            if self._seen[key] < self.capture_after:
                self.eager_calls += 1
                return self._eager(pv, pm)
        if entry is None:
            try:
                entry = self._capture(pv, pm)
            except Exception as e:  # capture unsupported by some library call: run eagerly, same kernels
                if self.strict:
                    raise
                self.capture_error = repr(e)
                self.enabled = False
                self.graphed = False
                torch.cuda.synchronize()
                return self._eager(pv, pm)
            self._entries[key] = entry
            while len(self._entries) > self.max_graphs:
                self._entries.popitem(last=False)
                self.evictions += 1
        else:
            self._entries.move_to_end(key)
        entry.static_in[0].copy_(pv)
        entry.static_in[1].copy_(pm)
        entry.graph.replay()
        entry.replays += 1
        if self._calls % self.status_every == 0:
            self._poll_decoder_status(pv.device)
        return entry.static_out


def configure_optimizers(model, lr=2e-6, lr_backbone=2e-7, lr_initialized=2e-4, weight_decay=1e-4,
                         initialized_keys=()):
    """The reference's three AdamW parameter groups (train_egtr.py:426-467)."""
    diff = ["backbone", "reference_points", "sampling_offsets"]
    init = list(initialized_keys) if lr_initialized is not None else []
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if not any(d in n for d in diff) and not any(d in n for d in init)]},
              {"params": [p for n, p in named if any(d in n for d in diff)], "lr": lr_backbone}]
    if init:
        groups.append({"params": [p for n, p in named if any(d in n for d in init)], "lr": lr_initialized})
    # same update rule as the reference's torch.optim.AdamW; on the GPU the single-kernel-per-group ("fused")
    # implementation replaces ~10 multi-tensor launches per group (the step is launch-bound: 42.5 M parameters)
    on_gpu = bool(named) and all(p.is_cuda for _, p in named)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, fused=True if on_gpu else None)


def init_distributed():
    """One process per GPU; RCCL over xGMI when GPUs are present, gloo on CPU (tests).  Reads the torchrun env."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.is_available():
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group("gloo")
    return world


class DataParallelTrainer:
    """Data-parallel train step with the reference's semantics (train_egtr.py:303-319, 770-779):
    identical replicas, per-rank micro-batches, gradient all-reduce (bucketed, overlapped with backward by DDP)
    once per optimizer step -- micro-steps before the accumulation boundary run under ``no_sync`` -- then
    clip-grad-norm 0.1 and AdamW.  ``num_boxes`` stays per-rank (model/egtr.py:976-980)."""

    def __init__(self, model, optimizer=None, accumulate=2, clip=0.1, bucket_cap_mb=25, graph=False, force_ddp=False,
                 grad_compression=None):
        """graph=True (single process, GPU, fixed image size): the static-shape part of the step -- backbone, encoder,
        decoder, detection + relation heads, forward AND backward -- is captured once into two HIP graphs
        (torch.cuda.make_graphed_callables over ``model.forward_tensors``) and replayed; the Hungarian matcher and the
        loss stay eager in between (they have data-dependent shapes).  Measured SLOWER than the eager step: the
        untraced eager step is GPU-bound with the host 25-40 ms ahead of the device, so a replay has no launch gaps to close,
        and it adds copies out of the graphs' static buffers (forward 15.1 vs 14.2 ms, backward 35.4 vs 29.9 ms).  Opt-in."""
        """``grad_compression``: None (fp32 all-reduce, the reference's DDP), "bf16" or "fp16" -- DDP's communication hook that
        casts every gradient bucket to 16 bits for the all-reduce and back (torch's ``bf16_compress_hook`` / ``fp16_compress_hook``):
        half the bytes per link for the bandwidth-bound stress configuration (SURVEY 2b; 165 MB fp32 per step otherwise).  The
        averaged gradient is then rounded to 8 / 11 significant bits before clipping -- opt-in, not the parity configuration."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.raw = model
        self.grad_compression = None
        self._graph_wanted = bool(graph) and self.world == 1
        self._graphed = None
        self._graph_key = None
        # force_ddp: wrap even in a one-rank process group, so that DDP's reducer (bucket hooks on the autograd Functions
        # that launch through the C ABI, gradient_as_bucket_view) can be exercised on a single-GPU box
        if self.world > 1 or (force_ddp and dist.is_initialized()):
            ids = [torch.cuda.current_device()] if next(model.parameters()).is_cuda else None
            self.model = torch.nn.parallel.DistributedDataParallel(
                model, device_ids=ids, find_unused_parameters=False, bucket_cap_mb=bucket_cap_mb,
                gradient_as_bucket_view=True)
            if grad_compression is not None:
                from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
                hooks = {"bf16": default_hooks.bf16_compress_hook, "fp16": default_hooks.fp16_compress_hook}
                if grad_compression not in hooks:
                    raise ValueError(f"grad_compression must be None, 'bf16' or 'fp16', got {grad_compression!r}")
                self.model.register_comm_hook(state=None, hook=hooks[grad_compression])
                self.grad_compression = grad_compression
        else:
            self.model = model
        self.opt = optimizer if optimizer is not None else configure_optimizers(model)
        self.accumulate = accumulate
        self.clip = clip
        self._micro = 0

    def _graphed_body(self, pixel_values, pixel_mask):
        key = (tuple(pixel_values.shape), pixel_values.dtype, tuple(pixel_mask.shape), pixel_mask.dtype)
        if self._graphed is None or key != self._graph_key:
            raw = self.raw

            class _Body(torch.nn.Module):  # parameters are found through the wrapped model
                def __init__(self):
                    super().__init__()
                    self.m = raw

                def forward(self, pv, pm):
                    return self.m.forward_tensors(pv, pm)

            # gradients accumulated by earlier micro-steps of this window must survive the warm-up / capture backward
            # passes (a shape change can fall on any micro-step when accumulate > 1): stash, capture, restore
            params = [p for p in raw.parameters()]
            stash = [p.grad for p in params]
            for p in params:
                p.grad = None
            self._graphed = torch.cuda.make_graphed_callables(
                _Body(), (pixel_values.detach().clone(), pixel_mask.detach().clone()), num_warmup_iters=3)
            self._graph_key = key
            for p, g in zip(params, stash):  # drops what the warm-up / capture backward passes left behind
                p.grad = g
        return self._graphed(pixel_values, pixel_mask)

    def common_step(self, batch):
        pv, pm = batch["pixel_values"], batch["pixel_mask"]
        if self._graph_wanted and pv.is_cuda and self.raw.training:
            return self.raw.loss_from_tensors(self._graphed_body(pv, pm), batch["labels"])
        out = self.model(pixel_values=pv, pixel_mask=pm, labels=batch["labels"],
                         output_attentions=False, output_attention_states=True, output_hidden_states=True)
        return out.loss, out.loss_dict

    def _refused_flag(self):
        """0-dim float32 on the device: 1 if the device matcher refused a cost matrix (NaN / -inf entries, infeasible) in any
        forward since the last optimizer step, on any rank; None when no device matcher ran.  No host synchronisation."""
        from .deformable_detr import DeformableDetrHungarianMatcher
        statuses = DeformableDetrHungarianMatcher.take_step_statuses()
        if not statuses:
            return None
        # the worst solver status code of the window (0 = assigned, 1 = NaN / -inf entries, 2 = infeasible) ...
        worst = torch.cat([s.reshape(-1) for s in statuses]).max().to(torch.float32).reshape(())
        if self.world > 1:
            # replicas must skip together (4 bytes) -- and RAISE together, with the RIGHT message: the MAX of the status CODES
            # is queued for raise_if_invalid on every rank (a 0 / 1 flag made every remote refusal read "invalid numeric
            # entries"); otherwise only the refusing rank would stop at its next step and the others would block
            # in the next gradient all-reduce
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            if worst.is_cuda:
                DeformableDetrHungarianMatcher.defer_status(worst)
        # ... and the 0 / 1 flag torch's GradScaler hands to the fused optimizers as `found_inf` (0-dim float32)
        return worst.ne(0).to(torch.float32).reshape(())

    def training_step(self, batch):
        """One micro-batch; returns (loss, loss_dict, stepped).

        A cost matrix the device matcher refuses (NaN / -inf: the reference raises scipy's ValueError inside the step,
        before backward) makes the loss and every gradient NaN.  The optimizer step of such a window is SKIPPED on the
        device (the fused AdamW's ``found_inf`` operand -- the mechanism of torch's GradScaler), so the weights and the
        optimizer state stay intact; the ValueError itself is raised by ``raise_if_invalid`` at the top of the next step
        (or by ``finalize()`` after the last one), without a host synchronisation inside the step that produced it.
        Optimizers without a device-side skip flag wait for the recorded status copy instead (one host wait per step)."""
        from .deformable_detr import DeformableDetrHungarianMatcher
        DeformableDetrHungarianMatcher.raise_if_invalid()
        self._micro += 1
        boundary = self._micro % self.accumulate == 0
        wrapped = self.model is not self.raw
        ctx = contextlib.nullcontext() if (boundary or not wrapped) else self.model.no_sync()
        with ctx:
            loss, loss_dict = self.common_step(batch)
            (loss / self.accumulate).backward()
        if boundary:
            refused = self._refused_flag()
            torch.nn.utils.clip_grad_norm_(self.raw.parameters(), self.clip)
            skip_on_device = refused is not None and all(g.get("fused") for g in self.opt.param_groups)
            if refused is not None and not skip_on_device:
                # no device-side skip in this optimizer: wait for the status copies and raise the matcher's ValueError now,
                # before the step (weights still intact); `refused` also covers the other ranks
                try:
                    DeformableDetrHungarianMatcher.raise_if_invalid()
                    if bool(refused.item()):
                        raise ValueError("matrix contains invalid numeric entries")
                except ValueError:
                    self.opt.zero_grad(set_to_none=True)
                    raise
            if skip_on_device:
                self.opt.found_inf = refused
            try:
                self.opt.step()
            finally:
                if skip_on_device:
                    del self.opt.found_inf
            self.opt.zero_grad(set_to_none=True)
        return loss.detach(), loss_dict, boundary

    def finalize(self):
        """After the last step of a run: raise what the device matcher refused in it (``training_step`` reports a refusal at
        the top of the NEXT step; there is none after the last one)."""
        from .deformable_detr import DeformableDetrHungarianMatcher
        DeformableDetrHungarianMatcher.raise_if_invalid()


def enable_gemm_tuning(results_file=None):
    """Let PyTorch's TunableOp pick, per GEMM shape, the fastest rocBLAS / hipBLASLt solution the first time the shape
    is seen (the MI355X analogue of the reference's ``torch.backends.cudnn.benchmark``-style autotuning).  The token
    sized encoder GEMMs (12 537 x 256 x {256, 384, 1024}) gain ~20 % over the libraries' default heuristics.  Call
    before the first forward; shapes must have been seen eagerly before a HIP-graph capture."""
    import os
    import tempfile
    try:
        import torch.cuda.tunable as tunable
    except Exception:  # pragma: no cover - very old torch
        return False
    tunable.enable(True)
    tunable.tuning_enable(True)
    if results_file is None:
        # one results file per PROCESS (= per rank): eight ranks tuning concurrently must not share a csv
        results_file = os.path.join(tempfile.gettempdir(),
                                    f"egtr_tunableop_r{os.environ.get('RANK', '0')}_{os.getpid()}.csv")
    tunable.set_filename(results_file, insert_device_ordinal=True)
    return True


def enable_conv_tuning():
    """MIOpen find mode (``torch.backends.cudnn.benchmark = True``): every convolution shape of the backbone gets the
    solver that measures fastest the first time it is seen, instead of the heuristic choice (ResNet-50 at 600x1000, fp32
    inference: 250 -> 261 images/s end to end).  Call before the first forward; shapes must have been seen eagerly
    before a HIP-graph capture.  No effect on the train step."""
    private_miopen_db()
    torch.backends.cudnn.benchmark = True
    return True


def private_miopen_db():
    """Multi-rank runs: give every rank its own MIOpen user database directory (find mode WRITES the tuned solvers
    there; eight ranks appending to one sqlite file race).  Must run before the first convolution; no-op for one rank
    or when the user already chose a path."""
    import tempfile
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "MIOPEN_USER_DB_PATH" not in os.environ:
        d = os.path.join(tempfile.gettempdir(), f"egtr_miopen_r{os.environ.get('RANK', '0')}")
        os.makedirs(d, exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = d
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", d)


@torch.no_grad()
def calculate_fps(model, batches, warmup=3, graphed=True, max_graphs=16, forward=None, capture_after=1):
    """evaluate_egtr.py:26-36 with warm-up and synchronisation (the reference's loop has neither).

    ``graphed=True`` (GPU): every batch goes through a ``GraphedForward`` -- one captured HIP graph per distinct image
    shape, LRU of ``max_graphs`` -- so a dataloader of differently sized images (evaluate_egtr.py:165-175: short side
    ``--min_size``, long side <= ``--max_size``) runs at graph-replay speed once each shape has been seen.  When ``batches``
    is a sequence its distinct shapes are captured BEFORE the clock starts (their two eager warm-up runs + capture are the
    analogue of the reference's cudnn autotuning on a new shape); for a plain iterator the first occurrence of a shape is
    inside the timed region.  ``forward``: an existing ``GraphedForward`` to reuse.  ``graphed=False``: eager launches.
    A real evaluation set has hundreds of distinct shapes: raise ``max_graphs`` (a 600x1000 graph holds < 0.5 GB of the 288 GB)
    and / or ``capture_after`` (eager until a shape comes back) accordingly."""
    import time
    model.eval()
    fwd = forward
    if fwd is None and graphed and torch.cuda.is_available():
        fwd = GraphedForward(model, enabled=True, strict=False, max_graphs=max_graphs, capture_after=capture_after)

    def run(batch):
        pv, pm = batch["pixel_values"].cuda(non_blocking=True), batch["pixel_mask"].cuda(non_blocking=True)
        if fwd is not None:
            return fwd(pv, pm)
        return model(pixel_values=pv, pixel_mask=pm, output_attentions=False, output_attention_states=True,
                     output_hidden_states=True)

    if fwd is not None and isinstance(batches, (list, tuple)):
        seen = set()
        for batch in batches:
            key = (tuple(batch["pixel_values"].shape), tuple(batch["pixel_mask"].shape))
            if key not in seen and len(seen) < fwd.max_graphs:
                seen.add(key)
                run(batch)
    n = 0
    t0 = None
    for i, batch in enumerate(batches):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        run(batch)
        if i >= warmup:
            n += batch["pixel_values"].shape[0]
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0) if t0 is not None and n else float("nan")


@torch.no_grad()
def triplet_candidates(outputs, num_labels, orig_sizes, max_topk=100, mode="multiple"):
    """The evaluator inputs of the reference's ``evaluate_batch`` (train_egtr.py:54-175) computed where the model outputs
    live.  Per image: object scores / classes (softmax over the first ``num_labels`` logits), boxes rescaled to the original
    image size, and -- by ``mode`` --
      * "multiple" (train_egtr.py:85-106, multiple-predicate evaluator): the ``max_topk`` best (subject, object, predicate)
        triplets by ``pred_rel * pred_connectivity * score_s * score_o`` (self-pairs excluded): ``pred_rel_inds`` [k, 3],
        ``rel_scores`` [k];
      * "single" (train_egtr.py:120-139, single-predicate evaluator): the ``max_topk`` best (subject, object) PAIRS by
        ``max_p(pred_rel * pred_connectivity) * score_s * score_o``: ``pred_rel_inds`` [k, 2], ``rel_scores`` [k, R] (the
        pair's whole predicate row);
      * "oi" (train_egtr.py:154-174, Open Images evaluator): every pair, ``sbj_obj_inds`` [N*N, 2] (cartesian product in
        row-major order) and ``pred_scores`` [N*N, R].

    The reference copies ``pred_rel`` [N, N, R] (8 MB at N = 200) to the host and runs a full numpy argsort over its
    entries per image (lib/pytorch_misc.py:27-34); here it is one batched ``topk`` on the device and only ``max_topk`` rows
    leave it.  Ties between equal scores may be ordered differently from numpy's argsort (both orders are valid argsorts).
    ``orig_sizes``: [B, 2] (h, w).  Returns a list of dicts of tensors on the outputs' device with the reference's
    ``pred_entry`` keys (+ ``triplet_scores`` for the two top-k modes)."""
    if mode not in ("multiple", "single", "oi"):
        raise ValueError(f"mode must be 'multiple', 'single' or 'oi', got {mode!r}")
    logits, boxes = outputs["logits"], outputs["pred_boxes"]
    rel = torch.clamp(outputs["pred_rel"], 0.0, 1.0)
    if "pred_connectivity" in outputs and outputs["pred_connectivity"] is not None:
        rel = rel * torch.clamp(outputs["pred_connectivity"], 0.0, 1.0)
    B, N, _, R = rel.shape
    obj_scores, pred_classes = torch.max(logits.softmax(-1)[..., :num_labels], -1)         # [B, N]
    sizes = torch.as_tensor(orig_sizes, dtype=torch.float32, device=rel.device).reshape(B, 2)
    cx, cy, w, h = boxes.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)
    scale = torch.stack([sizes[:, 1], sizes[:, 0], sizes[:, 1], sizes[:, 0]], -1)            # (w, h, w, h)
    xyxy = xyxy * scale[:, None, :]
    common = [{"pred_boxes": xyxy[b], "pred_classes": pred_classes[b], "obj_scores": obj_scores[b]} for b in range(B)]
    if mode == "oi":
        ar = torch.arange(N, device=rel.device)
        pairs = torch.cartesian_prod(ar, ar)
        return [dict(c, sbj_obj_inds=pairs, pred_scores=rel[b].reshape(N * N, R)) for b, c in enumerate(common)]
    sub_ob = obj_scores[:, :, None] * obj_scores[:, None, :]
    sub_ob = sub_ob.masked_fill(torch.eye(N, dtype=torch.bool, device=rel.device)[None], 0.0)
    if mode == "single":
        scores = (rel.max(-1)[0] * sub_ob).reshape(B, -1)                                     # [B, N*N]
        k = min(max_topk, scores.shape[1])
        top, flat = torch.topk(scores, k, dim=1)
        s_idx = torch.div(flat, N, rounding_mode="floor")
        o_idx = flat % N
        rows = rel.reshape(B, N * N, R).gather(1, flat[:, :, None].expand(-1, -1, R))         # [B, k, R]
        return [dict(c, pred_rel_inds=torch.stack([s_idx[b], o_idx[b]], -1), rel_scores=rows[b], triplet_scores=top[b])
                for b, c in enumerate(common)]
    scores = (rel * sub_ob.unsqueeze(-1)).reshape(B, -1)
    k = min(max_topk, scores.shape[1])
    top, flat = torch.topk(scores, k, dim=1)                                                  # sorted descending
    s_idx = torch.div(flat, N * R, rounding_mode="floor")
    o_idx = torch.div(flat, R, rounding_mode="floor") % N
    p_idx = flat % R
    rel_scores = rel.reshape(B, -1).gather(1, flat)
    return [dict(c, pred_rel_inds=torch.stack([s_idx[b], o_idx[b], p_idx[b]], -1), rel_scores=rel_scores[b],
                 triplet_scores=top[b]) for b, c in enumerate(common)]
