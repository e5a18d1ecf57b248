"""Data-parallel gradient exchange: one flat fp32 bucket, one RCCL all-reduce per optimizer step.

The reference scales with nn.DataParallel (train_partseg_shapenet.py:248-250): one process, one
thread per GPU, parameter broadcast + gradient reduce every iteration, per-replica BatchNorm
statistics.  Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), shapes
sharded by rank, and the 1.76 M parameters' gradients live in ONE contiguous buffer (7 MB) that is
all-reduced once -- the exchange is latency-bound on xGMI, so a single collective is the right
granularity (no per-layer buckets).  BatchNorm statistics stay per-rank (no SyncBN), as in the
reference; `sync_buffers` broadcasts rank 0's running stats ("replica 0 wins" in DataParallel).
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    """Per step: `zero()` before backward, `allreduce()` after it, then `optimizer.step()`.

    Gradients are produced by autograd as ordinary per-parameter tensors (zero() only drops the old
    ones, so backward *writes* instead of accumulating); `allreduce()` packs them into the flat fp32
    bucket with one multi-tensor copy, runs ONE all-reduce, scales by 1/world and points every
    `p.grad` at its slice of the bucket.  Without an initialised process group nothing is exchanged.

    Which parameters the optimizer sees is the reference's (torch 1.6 `optimizer.zero_grad()`,
    train_partseg_shapenet.py:383,436, zero-FILLS existing gradients): a parameter that has received a
    gradient in ANY earlier step keeps a (zero) gradient in the steps where autograd produces none for
    it -- `extra_conv_emb` in supervised steps, `conv2` in self-supervised ones -- so Adam still applies
    its weight decay and stale moments there; a parameter that never had one stays `None` and is
    skipped.  The same rule holds with 1 and with N ranks, so the trajectories agree."""

    def __init__(self, module, process_group=None, native=None, strict_seen=False, deferred_check=None):
        self.module = module
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # the bucket ends with one has-gradient flag per parameter: summed by the same all-reduce, they make "has this
        # parameter ever had a gradient" (self.seen) a GLOBAL fact -- with rank-local flags a rank whose shard produced
        # no gradient for a parameter (a shape batch with no cluster, a data-dependent branch) would skip it in Adam
        # while the others apply weight decay and moments to the averaged gradient: silent divergence of the replicas
        self.nflat = n
        self.flat = torch.zeros(n + len(self.params), dtype=torch.float32, device=dev)
        self.flags = self.flat[n:]
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.seen = [False] * len(self.params)   # has this parameter ever had a gradient ON ANY RANK (see the class docstring)
        self._grads = None
        if strict_seen and deferred_check:
            raise ValueError("FlatGradBucket: strict_seen=True asks for the synchronous, exact adoption of other ranks' "
                             "gradients; deferred_check=True for the check one exchange late -- choose one")
        self.strict_seen = strict_seen
        # deferred_check: the reduced flags are read one exchange late (no host read-back in the step).  Default: device
        # buckets unless strict_seen; True forces the protocol on a host bucket (the gloo tests drive it that way)
        self.deferred = (dev.type == "cuda" and not strict_seen) if deferred_check is None else bool(deferred_check)
        self._pending, self._flag_host = None, None
        self.dist = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.dist else 1
        # native=True / PRIFIT_NATIVE_RCCL=1: the all-reduce goes through the library's own RCCL communicator
        # (prifit_allreduce_flat, include/prifit_hip.h) instead of torch.distributed; device buckets only
        if native is None:
            import os
            native = os.environ.get("PRIFIT_NATIVE_RCCL", "0") == "1"
        self.native = None
        if native and self.dist and dev.type == "cuda":
            from .rccl import NativeComm
            self.native = NativeComm.from_process_group(process_group)

    def zero(self):
        self._grads = None
        for p in self.params:
            p.grad = None

    def grads(self):
        """The gradients as the optimizer will see them, in parameter order (None = skipped), valid after allreduce() until
        the next zero(): `optimizer.step(grads=bucket.grads())` saves FlatAdam a second pass over ~150 `p.grad` getters."""
        if self._grads is None:
            self._grads = [p.grad for p in self.params]
        return self._grads

    def _mark_seen(self):
        for i, p in enumerate(self.params):
            if p.grad is not None:
                self.seen[i] = True

    def pack(self, adopt=True):
        """Copy the per-parameter gradients into the bucket (zeros for parameters that got none) and set the
        has-gradient flags behind them; adopt: point `p.grad` of every parameter that has ever had a gradient at its
        slice (allreduce() does that after the exchange instead, with the other ranks' flags)."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        if len(have) != len(self.params):
            self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        local = [p.grad is not None for p in self.params]
        if all(local):
            self.flags.fill_(1.0)
        else:
            self.flags.copy_(torch.tensor(local, dtype=torch.float32), non_blocking=True)
        if adopt:
            self.seen = [s or l for s, l in zip(self.seen, local)]
            for v, p, s in zip(self.views, self.params, self.seen):
                p.grad = v if s else None
        return local

    def _adopt(self, local):
        """After the exchange: a parameter that had a gradient on ANY rank, now or earlier, gets its bucket slice.
        Whether another rank had a gradient this rank lacks is in the reduced flags, which are THE SAME on every rank, and
        so is `self.seen` before the exchange -- every decision below is therefore taken by all ranks alike.  The flags
        matter only while some parameter has never had a gradient anywhere; then they are read
        * synchronously (`strict_seen`, host buckets): a parameter with a gradient on some rank is adopted by all, exact;
        * or one exchange late (device buckets by default: no host read-back inside the step).  The step carries on with
          the rank-local answer, which is right whenever the ranks agree (reduced flag 0 or `world`); a flag strictly
          between marks a parameter some ranks stepped and others skipped, and EVERY rank raises at its next exchange
          (or in `flush()`, which the trainer and the benchmark call after their last step) -- all ranks see the same
          flags, so no rank is left waiting in a collective and a divergence of the replicas is never silent."""
        self._check_pending()
        unseen = [i for i, s in enumerate(self.seen) if not s]      # global knowledge: the same list on every rank
        self.seen = [s or l for s, l in zip(self.seen, local)]
        if unseen and not self.deferred:
            got = (self.flags > 0).tolist()
            self.seen = [s or g for s, g in zip(self.seen, got)]
        elif unseen:
            if self.flat.is_cuda:
                if self._flag_host is None:
                    self._flag_host = torch.empty(len(self.params), dtype=torch.float32).pin_memory()
                self._flag_host.copy_(self.flags, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (ev, unseen, self._flag_host)
            else:
                self._pending = (None, unseen, self.flags.clone())
        for v, p, s in zip(self.views, self.params, self.seen):
            p.grad = v if s else None

    def _check_pending(self):
        if self._pending is None:
            return
        ev, unseen, host = self._pending
        self._pending = None
        if ev is not None:
            ev.synchronize()
        bad = [i for i in unseen if 0.5 < float(host[i]) < self.world - 0.5]
        if bad:
            names = {id(p): n for n, p in self.module.named_parameters()}
            raise RuntimeError("gradient exchange: %s received a gradient on some ranks only (%s of %d); those ranks "
                               "stepped it, the others skipped it, and the replicas have diverged.  Every rank raises "
                               "this in the same exchange.  Construct FlatGradBucket(..., strict_seen=True) when ranks can "
                               "disagree on which parameters get gradients."
                               % (", ".join(names.get(id(self.params[i]), "#%d" % i) for i in bad),
                                  "/".join("%d" % round(float(host[i])) for i in bad), self.world))

    def flush(self):
        """Check the flags of the LAST exchange (deferred protocol): call after the final optimizer step."""
        self._check_pending()

    def allreduce(self):
        """Average gradients over ranks: one sum all-reduce of the flat bucket, then scale by 1/world.
        Single process: nothing moves; parameters that got no gradient this step but had one before get their
        (zeroed) bucket slice as gradient."""
        if not self.dist:
            grads = [p.grad for p in self.params]      # ONE pass over the `.grad` getters (~1 us each, 144 parameters)
            have = [g is not None for g in grads]
            if have != self.seen:                      # (steady state: equal lists, nothing else to do)
                self.seen = [s or h for s, h in zip(self.seen, have)]
                missing = [i for i, (s, h) in enumerate(zip(self.seen, have)) if s and not h]
                if missing:
                    torch._foreach_zero_([self.views[i] for i in missing])
                    for i in missing:
                        self.params[i].grad = grads[i] = self.views[i]
            self._grads = grads
            return
        local = self.pack(adopt=False)
        if self.native is not None:
            self.native.allreduce_(self.flat)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self._adopt(local)
        self._grads = None
        self.flat[:self.nflat].mul_(1.0 / self.world)

    def broadcast_parameters(self, src=0):
        if self.dist:
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=src, group=self.group)

    def sync_buffers(self, src=0):
        if self.dist:
            for b in self.module.buffers():
                dist.broadcast(b.data, src=src, group=self.group)
