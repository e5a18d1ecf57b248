"""Phase timing of the persistent tiled GEMM from in-kernel s_memtime stamps (DIAGNOSIS build only).

    touch prifit_amd/csrc/gemm.hip; PRIFIT_BUILD_DEFS=-DPERS_STAMPS python -m prifit_amd.build
    python tools/pers_stamps.py        (GPU box; rebuild without the flag afterwards)

Wave 0 of eight workgroups stamps its first eight tiles: 0 tile top | 1 all but the last k-tile done | 2 k-loop done |
3 epilogue stores issued | 4 statistics written | 5 next tile's first k-tile staged."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from prifit_amd import nn_ops
from prifit_amd.nn_ops import dll

NT, NN = 0, 1
def run(name, layout, M, N, K, affine, stats):
    A = torch.randn(M, K, device="cuda")
    Bm = torch.randn(N, K, device="cuda") if layout == NT else torch.randn(K, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    aff = (torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda")) if affine else None
    bias = torch.randn(N, device="cuda") if stats else None
    st = torch.empty(nn_ops.gemm_stats_slabs(M, N, K), 2, N, device="cuda") if stats else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(4):
        if i == 3: ev[0].record()
        nn_ops.gemm(layout, M, N, K, A, K, Bm, K if layout == NT else N, C, N, a_affine=aff, bias=bias, stats=st)
    ev[1].record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert dll().prifit_debug_pers_stamps(buf, 512) == 0
    t = np.array(buf, dtype=np.uint64).reshape(8, 8, 8).astype(np.int64)
    names = ["k-tiles but last", "last k-tile", "epilogue", "statistics", "stage next"]
    print("%s [%d x %d x %d]: %.0f us" % (name, M, N, K, 1e3 * ev[0].elapsed_time(ev[1])))
    for slot in range(0, 8, 2):
        d = np.diff(t[slot, :, :6], axis=1)[1:7]
        tile = t[slot, 2:8, 0] - t[slot, 1:7, 0]
        print("  slot %d  tile %6d | " % (slot, np.median(tile)) + " | ".join("%s %5d" % (n, v) for n, v in zip(names, np.median(d, axis=0))))

for ktail in (1,):
    if hasattr(dll(), "prifit_debug_pers_ktail"):
        torch.cuda.synchronize(); dll().prifit_debug_pers_ktail(ktail); print("-- skip empty k groups of the last k-tile:", ktail)
    run("NT forward (affine, bias, statistics)", NT, 393216, 256, 196, True, True)
    run("NN plain", NN, 393216, 128, 196, False, False)
def run_red(M, N, K):
    from prifit_amd.nn_ops import call, ptr, cur_stream, _LL
    dY = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); G = torch.empty(M, N, device="cuda")
    Y = torch.randn(M, N, device="cuda")
    v = [torch.rand(N, device="cuda") + 0.5 for _ in range(4)]
    t = dll().prifit_gemm_stats_tile_m(M, N)
    slab = torch.empty((M + t - 1) // t, 2, N, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(4):
        if i == 3: ev[0].record()
        call("prifit_gemm_dgrad_bnred_f32", M, N, K, ptr(dY), _LL(K), ptr(W), _LL(N), ptr(G), _LL(N), ptr(Y), _LL(N),
             ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), ptr(slab), None, cur_stream())
    ev[1].record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert dll().prifit_debug_pers_stamps(buf, 512) == 0
    t = np.array(buf, dtype=np.uint64).reshape(8, 8, 8).astype(np.int64)
    names = ["k-tiles but last", "last k-tile", "epilogue", "statistics", "stage next"]
    print("NN + BatchNorm-backward partials [%d x %d x %d]: %.0f us" % (M, N, K, 1e3 * ev[0].elapsed_time(ev[1])))
    for slot in range(0, 8, 2):
        d = np.diff(t[slot, :, :6], axis=1)[1:7]
        tile = t[slot, 2:8, 0] - t[slot, 1:7, 0]
        print("  slot %d  tile %6d | " % (slot, np.median(tile)) + " | ".join("%s %5d" % (n, v) for n, v in zip(names, np.median(d, axis=0))))

run_red(393216, 196, 256)
run_red(393216, 128, 196)
run("NT forward (affine, bias, statistics)", NT, 393216, 196, 128, True, True)
run("NN plain", NN, 393216, 128, 256, False, False)
