"""ctypes binding of libprifit_hip.so (the C ABI declared in include/prifit_hip.h).

The product path has NO fallback: if the shared object is missing or a call returns a non-zero
code, a RuntimeError is raised.  PyTorch is used only to own device memory and streams.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PRIFIT_LIB: another build of the same library (A/B measurements of kernel variants, tools/ab_libs.sh) -- the product
# library in lib/ is never overwritten by a measurement
LIB_PATH = os.environ.get("PRIFIT_LIB") or os.path.join(_HERE, "lib", "libprifit_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "prifit_hip.h")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_D = ctypes.c_double
_LL = ctypes.c_longlong

_dll = None


def declared_symbols(header_path=HEADER_PATH):
    """Names of every `int prifit_*(...)` / `long long prifit_*(...)` entry point declared in the public header."""
    return sorted(_declared(header_path))


def _declared(header_path=HEADER_PATH):
    with open(header_path) as f:
        text = f.read()
    return {name: ret for ret, name in re.findall(r"^\s*(int|long long)\s+(prifit_\w+)\s*\(", text, flags=re.M)}


def _ctype_of(param):
    """ctypes type of one C parameter declaration of the public header: every pointer (device pointers, host arrays of pointers,
    descriptor structs, the stream) travels as void *; scalars by their C type."""
    p = param.strip()
    if p in ("void", ""):
        return None
    if "*" in p or "[" in p:
        return ctypes.c_void_p
    if re.search(r"\blong\s+long\b", p):
        return ctypes.c_ulonglong if "unsigned" in p else ctypes.c_longlong
    if re.search(r"\bdouble\b", p):
        return ctypes.c_double
    if re.search(r"\bfloat\b", p):
        return ctypes.c_float
    if re.search(r"\b(int|int32_t|unsigned|uint32_t)\b", p):
        return ctypes.c_uint if "unsigned" in p or "uint32_t" in p else ctypes.c_int
    raise RuntimeError("prifit_hip.h: cannot map parameter %r to a ctypes type" % param)


def _signatures(header_path=HEADER_PATH):
    """{entry point: [ctypes types of its parameters]} parsed from the declarations of the public header.  With argtypes set,
    Python ints (tensor.data_ptr(), 64-bit sizes) and floats are converted by ctypes itself: no per-argument wrapper objects on
    the launch path, and a 64-bit value can never be truncated to a C int."""
    with open(header_path) as f:
        text = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
    sigs = {}
    for m in re.finditer(r"^\s*(?:int|long long)\s+(prifit_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        params = [x for x in (q.strip() for q in m.group(2).replace("\n", " ").split(",")) if x]
        types = [_ctype_of(q) for q in params]
        sigs[m.group(1)] = [t for t in types if t is not None]
    return sigs


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libprifit_hip.so is not built (%s). Run `python -m prifit_amd.build` "
                "(needs hipcc); there is no CPU fallback." % LIB_PATH)
        _dll = ctypes.CDLL(LIB_PATH)
        sigs = _signatures()
        for name, ret in _declared().items():
            fn = getattr(_dll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = ctypes.c_int if ret == "int" else ctypes.c_longlong
            if name not in sigs:
                raise RuntimeError("prifit_hip.h: no parameter list parsed for %s" % name)
            fn.argtypes = sigs[name]
    return _dll


def ptr(t):
    """Device address of a tensor for a pointer parameter (None = NULL); a plain int: argtypes does the conversion."""
    if t is None:
        return None
    return t.data_ptr()


_query_cache = {}


def query(name, *args):
    """A pure function of integer arguments exported by the library (prifit_*_supported / _slabs / _workspace / _tile_m ...),
    memoised: the launch path asks the same ~30 questions about the same shapes every step."""
    key = (name,) + args
    v = _query_cache.get(key)
    if v is None:
        v = _query_cache[key] = getattr(dll(), name)(*args)
    return v


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def cur_stream(device=None):
    """hipStream_t of PyTorch's current stream on `device` (default: the current device).  Called once per kernel launch
    (~190 times per training step): torch.cuda.current_stream() builds a Stream object through three Python layers (~10 us,
    2 ms of host time per step); the raw accessor inductor uses costs a fraction of a microsecond."""
    if _raw_stream is not None and _get_device is not None:
        if device is None:
            idx = _get_device()
        elif isinstance(device, int):
            idx = device
        else:
            d = torch.device(device)
            idx = _get_device() if d.index is None else d.index
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def call(name, *args):
    """Invoke an entry point; raises on a non-zero return code."""
    rc = getattr(dll(), name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("prifit_amd ops need device tensors (HIP backend only, no CPU path)")


def cf(t):
    """contiguous fp32"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
