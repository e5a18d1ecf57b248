"""Adam over one flat parameter buffer: ONE launch per optimizer step (csrc/optim.hip, prifit_adam_flat).

    opt = FlatAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    ... backward ...; bucket.allreduce(); opt.step()

The reference builds `torch.optim.Adam(classifier.parameters(), lr, betas=(0.9, 0.999), eps=1e-08, weight_decay)`
(train_partseg_shapenet.py:252-259) and steps it at :398 / :451.  torch's fused implementation needs three launches for the
MSG network's 144 tensors plus a multi-tensor add for the step counters, and ~0.6-0.9 ms of host time per step
(`_init_group`, grouping by device and dtype); here the parameters are moved ONCE into a flat fp32 buffer (`p.data` becomes a
view of it: the module, its state_dict and checkpoints do not notice), the moments live in two more, and a step is one
launch plus a comparison of the gradients' addresses with the table uploaded earlier (the caching allocator hands a static
step the same blocks every time; a changed address costs one small asynchronous upload).

Semantics kept from torch: a parameter whose `.grad` is None is skipped entirely (no weight decay, no moment decay, its step
count stays); per-parameter step counts; L2 weight decay added to the gradient; `param_groups[0]["lr"]` may be changed
between steps (the trainer's schedule, train_partseg_shapenet.py:325-330); `state_dict()` / `load_state_dict()` speak
torch.optim.Adam's format, so `optimizer_state_dict` of a checkpoint written by either loads into the other."""
import ctypes

import torch

from ._lib import call, cur_stream, dll, ptr, query


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no parameters")
        dev = self.params[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise RuntimeError("FlatAdam needs fp32 parameters on one GPU (HIP backend only, no CPU path)")
        align = query("prifit_adam_flat_alignment")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                               # the module keeps its Parameter objects; their storage is the flat buffer
        n = len(self.params)
        self._off = torch.tensor(self.offsets, dtype=torch.int32, device=dev)
        self._len = torch.tensor([p.numel() for p in self.params], dtype=torch.int32, device=dev)
        self._steps = [torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)]
        self._cur = 0                                       # _steps[_cur] holds the current counts
        self._gtab = torch.zeros(n, dtype=torch.int64, device=dev)
        self._gtab_host = [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._gtab_ev = [None, None]
        self._up = 0
        self._cached = None
        self.uploads = 0                                    # how often the address table changed (diagnosis: ~1-3 per run)
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}]

    # ------------------------------------------------------------------ step
    def _check_storage(self):
        p0, pl = self.params[0], self.params[-1]
        es = self.flat.element_size()
        if (p0.data_ptr() != self.flat.data_ptr() + self.offsets[0] * es or
                pl.data_ptr() != self.flat.data_ptr() + self.offsets[-1] * es):
            raise RuntimeError("FlatAdam: a parameter no longer lives in the flat buffer (module.to() / a re-assigned "
                               "`.data` after the optimizer was built); build the optimizer after moving the model")

    def _grad_table(self, grads=None):
        if grads is None:
            grads = [p.grad for p in self.params]           # (~1 us each: FlatGradBucket hands its own list over, see step())
        ptrs = tuple([0 if g is None else g.data_ptr() for g in grads])
        if ptrs != self._cached:
            for p, g in zip(self.params, grads):            # (only when an address changed: the layout the kernel assumes)
                if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.shape != p.shape):
                    raise RuntimeError("FlatAdam: gradients must be contiguous fp32 tensors of their parameter's shape on its device")
            k = self._up
            self._up ^= 1
            if self._gtab_ev[k] is not None:
                self._gtab_ev[k].synchronize()              # the upload that last used this pinned buffer has run
            host = self._gtab_host[k]
            host.copy_(torch.tensor(ptrs, dtype=torch.int64))
            self._gtab.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._gtab_ev[k] = ev
            self._cached = ptrs
            self.uploads += 1

    @torch.no_grad()
    def step(self, skip=None, grads=None):
        """One Adam step over every parameter that has a gradient.  skip: optional int32 device tensor; non-zero makes the launch a
        no-op (a step whose result is being discarded).  grads: the gradients as a list in parameter order (None entries = no
        gradient), when the caller has just read them anyway (ddp.FlatGradBucket.grads())."""
        self._check_storage()
        self._grad_table(grads)
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        src, dst = self._steps[self._cur], self._steps[self._cur ^ 1]
        call("prifit_adam_flat", ptr(self.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self._gtab), ptr(self._off),
             ptr(self._len), len(self.params), ctypes.c_longlong(self.total), ptr(src), ptr(dst), ctypes.c_float(g["lr"]),
             ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(g["eps"]), ctypes.c_float(g["weight_decay"]), ptr(skip),
             cur_stream())
        self._cur ^= 1

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @property
    def state(self):
        """torch.optim.Optimizer.state: {parameter: {"step", "exp_avg", "exp_avg_sq"}} for every parameter that has stepped
        (copies: the flat buffers are the live state)."""
        return {self.params[i]: st for i, st in self.state_dict()["state"].items()}

    # ------------------------------------------------------------------ checkpoints (torch.optim.Adam's format)
    def state_dict(self):
        steps = self._steps[self._cur].cpu().tolist()
        state = {}
        for i, (p, o, t) in enumerate(zip(self.params, self.offsets, steps)):
            if t > 0:
                state[i] = {"step": torch.tensor(float(t)), "exp_avg": self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": g["betas"], "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: %d parameters in the checkpoint, %d here" % (len(ids), len(self.params)))
        g0 = groups[0]
        self.param_groups[0].update(lr=g0["lr"], betas=tuple(g0["betas"]), eps=g0["eps"], weight_decay=g0["weight_decay"])
        steps = [0] * len(self.params)
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for pos, pid in enumerate(ids):
            st = sd["state"].get(pid)
            if not st:
                continue
            p, o = self.params[pos], self.offsets[pos]
            steps[pos] = int(float(st["step"]))
            self.exp_avg[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
        self._steps[self._cur].copy_(torch.tensor(steps, dtype=torch.int32))
