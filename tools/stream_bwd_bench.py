"""Stand-alone timing of one streaming-shape layer's backward: the one-pass dA + dW kernel (prifit_gemm_stream_bwd_f32) against
the separate streaming dA / dW kernels, on the SA1 / SA2 shapes of the step."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from prifit_amd._lib import call, cur_stream, ptr, dll
LL = ctypes.c_longlong
CASES = [(1572864, 128, 96, 128), (1572864, 96, 64, 0), (786432, 128, 64, 64), (786432, 64, 64, 0), (196608, 128, 128, 0)]
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 10
# (round 4: 32-row tiles for the Cout = 128 shapes -- half the LDS, two workgroups per CU -- measured with this tool:
#  [1572864 x 128 x 96 pool] 929 -> 916 us, [786432 x 128 x 64 pool] 342 -> 291, [196608 x 128 x 128] 145 -> 157, [49152 x 128 x 128]
#  47 -> 57: a gain on one pooled shape only (14 us of the step); not kept)
CASES.append((49152, 128, 128, 0))
tot = [0.0, 0.0]
for P, Cout, Kin, pool_K in CASES:
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    Y, A, W = rnd(P, Cout), rnd(P, Kin), rnd(Cout, Kin)
    s, t, ca, cb, cd = [rnd(Cout) for _ in range(5)]
    s1, t1, mu1, is1 = [rnd(Kin) for _ in range(4)]
    Gp = torch.empty(P, Kin, device="cuda"); dW = torch.zeros(Cout, Kin, device="cuda")
    sl = torch.empty(dll().prifit_gemm_stream_slabs(P, Cout), 2, Kin, device="cuda")
    ws = torch.empty(dll().prifit_gemm_stream_tn_workspace(Cout, Kin, LL(P)), device="cuda")
    if pool_K:
        arg = torch.randint(0, pool_K, (P // pool_K, Cout), device="cuda", generator=g, dtype=torch.int32); T = rnd(P // pool_K, Cout)
        bias_dw = torch.mv(W.t(), cd); G = None
        def sep():
            call("prifit_gemm_stream_tn_pool_f32", Cout, Kin, LL(P), ptr(Y), LL(Cout), ptr(A), LL(Kin), ptr(dW), LL(Kin), ptr(s1), ptr(t1), ptr(arg), ptr(T), ptr(cb), ptr(cd), pool_K, ptr(ws), cur_stream())
            call("prifit_gemm_stream_dgrad_pool_f32", P, Kin, Cout, ptr(Y), LL(Cout), ptr(W), LL(Kin), ptr(Gp), LL(Kin), ptr(bias_dw), ptr(arg), ptr(T), ptr(cb), pool_K, ptr(A), LL(Kin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl), None, cur_stream())
    else:
        G = rnd(P, Cout); arg = T = None
        def sep():
            call("prifit_gemm_stream_tn_bn_f32", Cout, Kin, LL(P), ptr(G), ptr(Y), LL(Cout), ptr(A), LL(Kin), ptr(dW), LL(Kin), ptr(s1), ptr(t1), ptr(s), ptr(t), ptr(ca), ptr(cb), ptr(cd), ptr(ws), cur_stream())
            call("prifit_gemm_stream_dgrad_bn_f32", P, Kin, Cout, ptr(G), ptr(Y), LL(Cout), ptr(W), LL(Kin), ptr(Gp), LL(Kin), ptr(s), ptr(t), ptr(ca), ptr(cb), ptr(cd), ptr(A), LL(Kin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl), None, cur_stream())
    sl2 = torch.empty(dll().prifit_gemm_stream_bwd_slabs(LL(P), Cout, Kin), 2, Kin, device="cuda")
    ws2 = torch.empty(dll().prifit_gemm_stream_bwd_workspace(LL(P), Cout, Kin), device="cuda")
    def fused():
        call("prifit_gemm_stream_bwd_f32", LL(P), Cout, Kin, ptr(G), ptr(Y), ptr(None if pool_K else s), ptr(None if pool_K else t), ptr(None if pool_K else ca), ptr(cb), ptr(cd), ptr(arg), ptr(T), pool_K, ptr(W), LL(Kin), ptr(A), LL(Kin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(Gp), LL(Kin), ptr(sl2), ptr(dW), LL(Kin), ptr(ws2), None, cur_stream())
    dW.zero_(); sep(); torch.cuda.synchronize(); ref = (Gp.clone(), dW.clone())
    dW.zero_(); Gp.zero_(); fused(); torch.cuda.synchronize()
    eg = ((Gp - ref[0]).norm() / ref[0].norm()).item(); ew = ((dW - ref[1]).norm() / ref[1].norm()).item()
    assert eg < 1e-5 and ew < 1e-4, (eg, ew)
    a, b = timeit(sep), timeit(fused)
    tot[0] += a; tot[1] += b
    gb = 4.0 * P * ((1 if pool_K else 2) * Cout + 2 * Kin) / 1e9
    print("[%8d x %3d x %3d %s] separate %7.1f us   one pass %7.1f us  (%.0f GB/s, %.0f TFLOP/s)" % (P, Cout, Kin, "pool" if pool_K else "bn  ", a, b, gb / b * 1e6, 4.0 * P * Cout * Kin / b / 1e6))
print("total: separate %.1f us, one pass %.1f us" % tuple(tot))
