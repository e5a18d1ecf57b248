"""Import shim for the upstream reference (THIS CONTAINER ONLY).

TEST INFRASTRUCTURE.  Used only by oracle/make_golden.py and oracle/check_oracle.py
to (1) validate the CPU restatement in oracle/prifit_oracle.py and (2) emit the golden
fixtures committed under tests/golden/.  Nothing here travels to the GPU box in a form
that is needed at run time: /root/reference does not exist there, and no test, smoke()
or bench leg imports this module.

The reference is pure Python but hard-imports a few packages that are not installed
(open3d, trimesh, ipdb, transforms3d, lap, tensorboard_logger) and calls `.cuda()`
unconditionally (e.g. models/pointnet2_part_seg_msg.py:89, src/ellipsoid_fitting.py:38,
src/mean_shift.py:178).  We inject empty stub modules and make `.cuda()` the identity so
the reference's CPU arithmetic runs unmodified.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("PRIFIT_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


class _Anything:
    """Attribute sink: any attribute access / call returns another sink."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__all__ = list(attrs)
    for k, v in attrs.items():
        setattr(m, k, v)
    def _missing(n):
        if n.startswith("__"):      # inspect.getmodule() probes __file__ of every module in sys.modules
            raise AttributeError(n)
        return _Anything()

    m.__getattr__ = _missing  # type: ignore[attr-defined]
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Make `import models.pointnet_util`, `import convex_loss`, ... resolve to the reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _stub("open3d", utility=_Anything(), geometry=_Anything(), visualization=_Anything(), io=_Anything())
    _stub("trimesh")
    _stub("ipdb", set_trace=lambda *a, **k: None)
    _stub("transforms3d")
    _stub("transforms3d.affines", compose=_Anything())
    _stub("transforms3d.euler", euler2mat=_Anything())
    _stub("lap")
    _stub("tensorboard_logger")
    try:
        import matplotlib  # noqa: F401
    except Exception:  # pragma: no cover
        _stub("matplotlib")
        _stub("matplotlib.pyplot")
        _stub("mpl_toolkits")
        _stub("mpl_toolkits.mplot3d", Axes3D=_Anything())

    ident_t = lambda self, *a, **k: self
    torch.Tensor.cuda = ident_t  # type: ignore[assignment]
    torch.nn.Module.cuda = ident_t  # type: ignore[assignment]
    torch.get_device = lambda t: None  # src/mean_shift.py:178 passes it to .cuda()
    torch.Tensor.get_device = lambda self: None  # src/fitting_utils.py:70
    torch.cuda.empty_cache = lambda: None
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _installed = True


def ref(module: str):
    """Import a reference module (e.g. 'models.pointnet_util') and undo its global reseeding
    (src/fitting_utils.py:9-10 seeds torch and numpy to 2 at import time)."""
    install()
    st_t = torch.get_rng_state()
    st_n = np.random.get_state()
    m = importlib.import_module(module)
    torch.set_rng_state(st_t)
    np.random.set_state(st_n)
    return m
