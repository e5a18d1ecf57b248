"""The training iteration of the hot path: what
train_partseg_shapenet.py:252-259, :321-340, :372-399, :436-451, :467-475 does around the model, for one
process per GPU.

    tr = Trainer(model)                       # Adam(lr, betas .9/.999, eps 1e-8, weight_decay) as upstream :252-259
    tr.set_epoch(epoch)                       # lr / BatchNorm-momentum schedule, upstream :325-334
    loss, acc = tr.supervised_step(points, target)              # upstream :372-399
    ss = tr.selfsup_step(chamfer_points, quantile=.05, ...)     # upstream :436-451
    tr.save(path) / tr.load(path)             # checkpoint dict of upstream :467-475
    tr.finish()                               # REQUIRED after the last step under torch.distributed: the deferred
                                              # has-gradient check of the final exchange (ddp.FlatGradBucket.flush)

The augmentation (random per-shape scale in [0.8, 1.25] and shift in [-0.1, 0.1], provider.py:278-303) runs
on the device.  Gradients are exchanged with `FlatGradBucket` when torch.distributed is initialised."""
import numpy as np
import os

import torch
import torch.nn.functional as F

from .ddp import FlatGradBucket

LEARNING_RATE_CLIP = 1e-5      # upstream :285
MOMENTUM_ORIGINAL = 0.1        # upstream :286
MOMENTUM_DECAY = 0.5           # upstream :287


def random_scale_shift(points, generator=None, scale_low=0.8, scale_high=1.25, shift_range=0.1):
    """provider.py:278-303 on a [B,N,3+] device tensor (first three channels), one scale and shift per shape."""
    B = points.shape[0]
    dev = points.device
    scales = torch.empty(B, 1, 1, device=dev).uniform_(scale_low, scale_high, generator=generator)
    shifts = torch.empty(B, 1, 3, device=dev).uniform_(-shift_range, shift_range, generator=generator)
    out = points.clone()
    out[:, :, 0:3] = out[:, :, 0:3] * scales + shifts
    return out


class SpeculativeRunner:
    """Run a forward+backward closure with fit_ops.speculative() (no mid-step host read-back); if the clustering
    verdict, read once everything is enqueued, says a quantile-doubling retry was needed, put the BatchNorm
    buffers back, clear the gradients and run the closure again the synchronous way."""

    def __init__(self, model):
        # one list per dtype: _foreach_copy_ then runs as a single multi-tensor kernel per list instead of one
        # copy per buffer (BatchNorm statistics are fp32, the batch counters int64)
        groups = {}
        for b in model.buffers():
            groups.setdefault(b.dtype, []).append(b)
        self.bufs = list(groups.values())
        self.snap = [[torch.empty_like(b) for b in g] for g in self.bufs]
        self.model = model
        self.fallbacks = 0
        self.runs = 0
        # set by a caller that wants the NEXT run to go through the fall-back once whatever its verdict (bench.py's first
        # warm-up step: the first fall-back of a process grows the caching allocator by ~4 GB, 0.2 s of hipMalloc)
        self.force_next = False

    def run(self, fn, reset):
        from . import fit_ops

        for dst, src in zip(self.snap, self.bufs):
            torch._foreach_copy_(dst, src)
        # python-side state the forward mutates: the entropy weight schedule `beta *= 0.99` per convex-loss forward
        # (models/pointnet2_part_seg_msg.py:96-99) must advance once per ACCEPTED step, not once per attempt
        beta = getattr(self.model, "beta", None)
        rng = torch.cuda.get_rng_state() if torch.cuda.is_available() else None
        with fit_ops.speculative() as spec:
            out = fn()
        self.runs += 1
        forced, self.force_next = self.force_next, False
        if spec.ok() and not forced:
            return out
        self.fallbacks += 1
        for dst, src in zip(self.bufs, self.snap):
            torch._foreach_copy_(dst, src)
        if beta is not None:
            self.model.beta = beta
        if rng is not None:
            torch.cuda.set_rng_state(rng)   # dropout masks / random tables of the retried step = those of the first try
        reset()
        return fn()


class _Backbone(torch.nn.Module):
    """`net.embed_features` as a module of its own (what make_graphed_callables captures): parameters = the network's."""

    def __init__(self, net, fps_start):
        super().__init__()
        self.net, self.fps_start = net, fps_start

    def forward(self, xyz, cls_label):
        return self.net._embed_features_eager(xyz, cls_label, self.fps_start)


def graph_backbone(net, xyz, cls_label, fps_start):
    """Capture the backbone of `net` (everything up to `feat`: set abstraction, feature propagation, conv1 + bn1 --
    upstream models/pointnet2_part_seg_msg.py:64-88 -- forward AND backward) into two HIP graphs and install it as
    `net.embed_features`: ~300 of a step's ~440 launches then cost the host one graph launch each way.  The shapes are
    static and nothing in there reads back to the host, the kernels only enqueue on the current stream (include/
    prifit_hip.h), so the capture is `torch.cuda.make_graphed_callables` as is; the fit path stays eager (its
    cluster-count verdict is read on the host, SpeculativeRunner).  Static inputs: `xyz` / `cls_label` of these shapes
    (other tensors are copied in), `fps_start` = these very tensors.  BatchNorm buffers are put back after the warm-up
    iterations of the capture.  Returns the graphed callable; `net.embed_features` replays it in training mode with
    gradients enabled and falls back to the eager backbone (`net._embed_features_eager`) otherwise -- an eval / no_grad
    forward must not replay a training-mode graph (batch statistics, running-stat updates)."""
    from . import arena
    if not hasattr(net, "_embed_features_eager"):
        net._embed_features_eager = net.embed_features
    bufs = [b for b in net.buffers()]
    snap = [b.clone() for b in bufs]
    was_enabled = arena._ENABLED
    arena._ENABLED = False          # zero-filled tensors inside the graphs are the graphs' own (re-zeroed at every replay)
    try:
        graphed = torch.cuda.make_graphed_callables(_Backbone(net, fps_start), (xyz, cls_label), num_warmup_iters=2,
                                                    allow_unused_input=True)
    finally:
        arena._ENABLED = was_enabled
        with torch.no_grad():
            for b, s_ in zip(bufs, snap):
                b.copy_(s_)

    shape = (tuple(xyz.shape), tuple(cls_label.shape))

    def embed_features(xyz_, cls_, fps_start_=None):
        same_start = fps_start_ is None or fps_start_ is fps_start or (
            len(fps_start_) == len(fps_start) and all(a is b for a, b in zip(fps_start_, fps_start)))
        if (not net.training or not torch.is_grad_enabled() or (tuple(xyz_.shape), tuple(cls_.shape)) != shape
                or not same_start):
            return net._embed_features_eager(xyz_, cls_, fps_start_)
        return graphed(xyz_, cls_)

    net.embed_features = embed_features
    return graphed


class Trainer:
    def __init__(self, model, num_part=50, learning_rate=0.001, decay_rate=1e-4, lr_decay=0.5, step_size=20, lmbda=1.0,
                 fused_adam=True, strict_seen=False):
        # one process per GPU: this rank's own cores and capped CPU thread pools (prifit_amd/hostcfg.py; a single process is
        # left alone, and a launcher that already did it -- bench.py -- makes this a no-op)
        from . import hostcfg
        self.host = hostcfg.apply_from_env()
        self.model = model
        self.num_part = num_part
        self.lr0, self.lr_decay, self.step_size, self.lmbda = learning_rate, lr_decay, step_size, lmbda
        # one launch per step over a flat parameter buffer on the GPU (prifit_amd/optim.py; PRIFIT_FLAT_ADAM=0 / fused_adam=False:
        # torch's optimizer, the A/B arm); host tensors (the gloo tests of the exchange logic) take torch's
        on_gpu = next(model.parameters()).is_cuda
        if on_gpu and fused_adam and os.environ.get("PRIFIT_FLAT_ADAM", "1") != "0":
            from .optim import FlatAdam
            self.optimizer = FlatAdam(model.parameters(), lr=learning_rate, betas=(0.9, 0.999), eps=1e-08, weight_decay=decay_rate)
        else:
            self.optimizer = torch.optim.Adam(model.parameters(), lr=learning_rate, betas=(0.9, 0.999), eps=1e-08,
                                              weight_decay=decay_rate, fused=bool(fused_adam and on_gpu))
        # strict_seen: ranks may disagree on which parameters get a gradient (see ddp.FlatGradBucket._adopt)
        self.bucket = FlatGradBucket(model, strict_seen=strict_seen)
        self._next = None          # the batch prepared by prefetch_selfsup
        # DataParallel replicates module 0's parameters and buffers onto every GPU at each forward
        # (train_partseg_shapenet.py:248-250); with one process per GPU the ranks start from rank 0's model instead
        self.bucket.broadcast_parameters(0)
        self.speculative = SpeculativeRunner(model) if next(model.parameters()).is_cuda else None
        self.epoch = 0
        self.train_acc = 0.0

    # ------------------------------------------------------------------ schedule (upstream :325-334)
    def set_epoch(self, epoch):
        self.epoch = epoch
        lr = max(self.lr0 * (self.lr_decay ** (epoch // self.step_size)), LEARNING_RATE_CLIP)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        momentum = max(MOMENTUM_ORIGINAL * (MOMENTUM_DECAY ** (epoch // self.step_size)), 0.01)
        for m in self.model.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.momentum = momentum
        return lr, momentum

    def _apply(self, loss):
        if loss is not None:
            loss.backward()
        self.bucket.allreduce()
        if hasattr(self.optimizer, "exp_avg"):           # FlatAdam: hand over the gradient list the exchange just read
            self.optimizer.step(grads=self.bucket.grads())
        else:
            self.optimizer.step()

    # ------------------------------------------------------------------ supervised step (upstream :372-399)
    def supervised_step(self, points, target, category_label=None, augment=True, fps_start=None):
        """points [B,N,3(+3)] channels-last, target [B,N] int64 part labels."""
        B, N, _ = points.shape
        if augment:
            points = random_scale_shift(points)
        xyz = points.transpose(2, 1).contiguous()
        if category_label is None:
            category_label = torch.zeros(B, 1, 16, device=points.device)
        self.bucket.zero()
        self.model.train()
        out = self.model(xyz, category_label, include_convex_loss=False, fps_start=fps_start)
        seg_pred = out[0].contiguous().view(-1, self.num_part)
        tgt = target.view(-1)
        loss = F.cross_entropy(seg_pred, tgt)      # get_loss: CE on log-probabilities (upstream msg:137-144)
        self._apply(loss)
        with torch.no_grad():
            acc = (seg_pred.argmax(1) == tgt).float().mean()
        return loss.detach(), acc

    # ------------------------------------------------------------------ self-supervised step (upstream :436-451)
    def _selfsup_batch(self, chamfer_points, npoint, augment, subset):
        """Augmentation + the random `npoint`-subset that is the model input (upstream :439-441), channels-first."""
        B, M, _ = chamfer_points.shape
        if augment:
            chamfer_points = random_scale_shift(chamfer_points)
        cham = chamfer_points.transpose(2, 1).contiguous()
        if subset is None:
            subset = torch.from_numpy(np.random.choice(M, npoint, replace=False)).to(cham.device)
        return cham, cham[:, :, subset].contiguous()

    def prefetch_selfsup(self, chamfer_points, npoint=2048, augment=True, subset=None, fps_start=None):
        """Prepare the NEXT self-supervised batch now -- augmentation and subset, what a DataLoader worker does upstream --
        and start its farthest-point sampling behind the backbone forward of the step that runs next (`model.sample_ahead`
        through the `after_backbone` hook: the sampling chain depends on the coordinates alone, is 640 serial rounds on one
        workgroup per shape, and there it runs beside the matrix-bound mean-shift kernels instead of at the head of its
        own step; same indices as in-line sampling).  The following `selfsup_step()` WITHOUT `chamfer_points` consumes it.
        Models without `sample_ahead` (DGCNN) only get the batch prepared."""
        if self._next is not None:
            raise RuntimeError("prefetch_selfsup: the batch of an earlier prefetch has not been consumed by a "
                               "selfsup_step() yet -- a second prefetch would drop it (its augmentation and RNG draw spent)")
        cham, points = self._selfsup_batch(chamfer_points, npoint, augment, subset)
        nxt = {"cham": cham, "points": points, "ahead": None, "fps_start": fps_start}
        self._next = nxt
        if points.is_cuda and hasattr(self.model, "sample_ahead"):
            def hook():
                self.model.after_backbone = None
                if self._next is nxt and nxt["ahead"] is None:
                    nxt["ahead"] = self.model.sample_ahead(points, fps_start)
            self.model.after_backbone = hook

    def selfsup_step(self, chamfer_points=None, npoint=2048, quantile=0.01, msc_iterations=20, max_num_clusters=25,
                     augment=True, subset=None, **loss_kwargs):
        """chamfer_points [B,M,3]: the model input is a random `npoint`-subset of them (upstream :441).
        chamfer_points=None: the batch prepared by `prefetch_selfsup`."""
        if chamfer_points is None:
            if self._next is None:
                raise RuntimeError("selfsup_step() without chamfer_points needs a prefetch_selfsup(...) before it")
            nxt, self._next = self._next, None
            if getattr(self.model, "after_backbone", None) is not None and nxt["ahead"] is None:
                self.model.after_backbone = None      # no step ran in between: sample in line
            cham, points = nxt["cham"], nxt["points"]
            start = nxt["ahead"] if nxt["ahead"] is not None else nxt["fps_start"]
            if start is not None:
                loss_kwargs = dict(loss_kwargs, fps_start=start)
        else:
            cham, points = self._selfsup_batch(chamfer_points, npoint, augment, subset)
        B = cham.shape[0]
        category_label = torch.zeros(B, 1, 16, device=cham.device)
        self.bucket.zero()
        self.model.train()

        def fwd_bwd():
            out = self.model(points, category_label, chamfer_points=cham, include_convex_loss=True, quantile=quantile,
                             msc_iterations=msc_iterations, max_num_clusters=max_num_clusters, **loss_kwargs)
            ss_loss = torch.mean(out[3]) * self.lmbda
            ss_loss.backward()
            return ss_loss

        if self.speculative is not None:
            ss_loss = self.speculative.run(fwd_bwd, self.bucket.zero)
        else:
            ss_loss = fwd_bwd()
        self._apply(None)
        return ss_loss.detach()

    # ------------------------------------------------------------------ checkpoints (upstream :467-475, :263-274)
    def finish(self):
        """After the last optimizer step: the deferred has-gradient check of the final exchange (ddp.FlatGradBucket.flush)."""
        self.bucket.flush()

    def sync_buffers(self):
        """COLLECTIVE: rank 0's BatchNorm statistics win on every rank ("replica 0" of DataParallel).  Every rank enters it,
        so it also runs the deferred has-gradient check of the last exchange first: the documented pattern
        `sync_buffers(); if rank == 0: write_checkpoint()` can then not persist a model taken after a divergent step."""
        self.bucket.flush()
        self.bucket.sync_buffers(0)

    def write_checkpoint(self, path):
        """Local, no communication: the checkpoint dict of upstream :467-475 -- safe under `if rank == 0:`."""
        torch.save({"epoch": self.epoch, "train_acc": self.train_acc, "model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict()}, path)

    def save(self, path):
        """sync_buffers() + write_checkpoint() on rank 0.  COLLECTIVE under torch.distributed: EVERY rank must call it
        (a caller that guards it with `if rank == 0:` would leave rank 0 alone in the broadcast -- such a caller uses
        write_checkpoint, after a sync_buffers() that all ranks entered)."""
        import torch.distributed as dist
        self.bucket.flush()
        self.sync_buffers()
        if dist.is_initialized() and dist.get_rank() != 0:
            return
        self.write_checkpoint(path)

    def load(self, path):
        ck = torch.load(path, map_location="cpu")
        self.model.load_state_dict(ck["model_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self.epoch = ck["epoch"]
        self.bucket.broadcast_parameters(0)
        return ck
