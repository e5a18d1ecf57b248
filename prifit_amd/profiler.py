"""Per-kernel-family timing with HIP events on the launch stream (used by bench.py for the roofline).

When a section is enabled, every launch wrapped by `span(name, work)` is bracketed by two events on
PyTorch's current stream (the stream the C-ABI kernels are launched on); `work` is the algorithmic
cost of the launch (flops or bytes).  `collect()` synchronises and returns, per name,
(launches, total_ms, total_work).  Disabled (the default) it costs one dict lookup per launch.
"""
import contextlib

import torch

import os

_enabled = set()
_records = {}
SHAPES = os.environ.get("PRIFIT_SPAN_SHAPES", "0") == "1"   # diagnosis: callers append the problem shape to the span name


def tag(name, *dims):
    return name + ("[" + "x".join(str(d) for d in dims) + "]" if SHAPES else "")


def enable(*names):
    _enabled.update(names)


def disable():
    _enabled.clear()


def reset():
    _records.clear()


class _Null:
    """The disabled span: one shared object, nothing allocated per launch (a generator-based context manager costs ~2 us, and a
    training step opens ~100 spans)."""
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _Null()


class _Span:
    __slots__ = ("name", "work", "a", "b")

    def __init__(self, name, work):
        self.name, self.work = name, float(work)

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        _records.setdefault(self.name, []).append((self.a, self.b, self.work))
        return False


def span(name, work=0.0):
    if not _enabled or (name not in _enabled and "*" not in _enabled):
        return _NULL
    return _Span(name, work)


def collect():
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        ms = sum(a.elapsed_time(b) for a, b, _ in recs)
        out[name] = (len(recs), ms, sum(w for _, _, w in recs))
    return out
