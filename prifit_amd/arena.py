"""One zero-fill per step instead of one per zero-initialised tensor.

A training step allocates ~50 zero-initialised fp32 tensors (split-K / atomic accumulation targets, gradient slots of the
fit kernels, weight-gradient arenas ...): each `torch.zeros` is its own fill launch (102 launches and 0.34 ms per c3 step
in the round-1 profile).  `begin_step()` -- called by the model's forward -- allocates ONE zeroed buffer sized by the
previous step's demand; `zeros()` carves 256-byte aligned views out of it and falls back to `torch.zeros` when the pool
is absent, exhausted or of another device/dtype.  Views keep their buffer alive, so nothing is ever reused while a
tensor carved from it exists; a new step gets a new buffer.
"""
import os

import torch

_ALIGN = 64          # floats
_ENABLED = os.environ.get("PRIFIT_ZERO_POOL", "1") != "0"
_pool = None         # (buffer, [offset])
_demand = 0          # floats asked for since the last begin_step
_last_demand = 0


def begin_step(device):
    """Start a new pool (sized by what the previous step asked for)."""
    global _pool, _demand, _last_demand
    if not _ENABLED:
        return
    _last_demand = max(_demand, 0)
    _demand = 0
    size = _last_demand + _last_demand // 8          # head-room: shapes that vary a little from step to step
    _pool = (torch.zeros(size, dtype=torch.float32, device=device), [0]) if size else None


def zeros(*shape, device, dtype=torch.float32):
    """fp32 zeros of `shape` on `device`: a view of the step's pool when it fits."""
    global _demand
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    n = 1
    for s in shape:
        n *= int(s)
    if dtype != torch.float32 or not _ENABLED:
        return torch.zeros(shape, dtype=dtype, device=device)
    padded = (n + _ALIGN - 1) // _ALIGN * _ALIGN
    _demand += padded
    if _pool is not None:
        buf, off = _pool
        if buf.device == torch.device(device) and off[0] + padded <= buf.numel():
            out = buf[off[0]:off[0] + n].view(shape)
            off[0] += padded
            return out
    return torch.zeros(shape, dtype=dtype, device=device)


def zeros_like(t):
    return zeros(tuple(t.shape), device=t.device, dtype=t.dtype)
