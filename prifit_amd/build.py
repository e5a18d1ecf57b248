"""Builds libprifit_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m prifit_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object is git-ignored but travels with gpurun
snapshots.  Index-critical sources are compiled with -ffp-contract=off (bit-exact recipes).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libprifit_hip.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"

# source -> extra flags
SOURCES = {
    "pointops.hip": ["-ffp-contract=off"],
    "sa_group.hip": ["-ffp-contract=off"],
    "gemm.hip": [],
    "gemm_stream.hip": [],
    "gemm_stream_bwd.hip": [],
    "pool_alg.hip": [],
    "sa_gather_bwd.hip": [],
    "bn.hip": [],
    "meanshift.hip": [],
    "meanshift_fused.hip": [],
    "meanshift_rows.hip": [],
    "meanshift_split.hip": [],
    "fit.hip": [],
    "fit_glue.hip": [],
    "optim.hip": [],
    "edge_conv.hip": [],
    "dgcnn.hip": ["-ffp-contract=off"],
    "comm.hip": [],        # host-only: the RCCL export (RCCL itself is resolved with dlopen at run time)
}
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + os.path.join(ROOT, "include"),
          "-I" + CSRC, "-Wall", "-Wno-unused-function"] + os.environ.get("PRIFIT_BUILD_DEFS", "").split()  # diagnosis builds: -DTN_DEBUG=1 ...


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile stale objects and link the shared object.  Safe to call from several processes at once (the ranks of
    a torch.distributed.run job): the whole build runs under an exclusive file lock and the library is linked to a
    temporary name and renamed into place, so nobody ever dlopens a half-written file."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "prifit_hip.h"))
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        tmp = LIB + ".tmp.%d" % os.getpid()
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs + ["-ldl"])
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
