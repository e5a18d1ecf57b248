"""EXPERIMENT: the row-dense part of a pooled layer's backward in the algebraic form (prifit_pool_alg_dense_f32:
Gp = relu(bn(Yp)) M + v, the BatchNorm-backward sums of the layer below, the Gram matrix A^T A) -- checked against torch and
timed next to the streaming pair the step uses today for the same layer (dA + dW over Cout x Cin, reading the pooled layer's Y)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from prifit_amd._lib import call, cur_stream, ptr, dll
LL = ctypes.c_longlong
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for P, Cout, Cin, pool_K in [(1572864, 128, 96, 128), (786432, 128, 64, 64), (196608, 256, 128, 64)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    Yp, M, v = rnd(P, Cin), rnd(Cin, Cin) * 0.1, rnd(Cin)
    M = (M + M.t()).contiguous()
    s1, t1, mu1 = rnd(Cin), rnd(Cin), rnd(Cin)
    is1 = rnd(Cin).abs() + 0.5
    Gp = torch.empty(P, Cin, device="cuda"); gram = torch.empty(Cin, Cin, device="cuda")
    ns = dll().prifit_pool_alg_slabs(LL(P), Cin)
    sl = torch.empty(ns, 2, Cin, device="cuda"); ws = torch.empty(dll().prifit_pool_alg_workspace(LL(P), Cin), device="cuda")
    asum0 = torch.empty(Cin, device="cuda")
    def alg():
        call("prifit_pool_alg_dense_f32", LL(P), Cin, ptr(Yp), LL(Cin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(M), LL(Cin), ptr(v),
             ptr(Gp), LL(Cin), ptr(sl), ptr(gram), ptr(asum0), ptr(ws), cur_stream())
    alg(); torch.cuda.synchronize()
    n = min(P, 65536)
    A = torch.relu(Yp * s1 + t1)
    want = A[:n].double() @ M.double() + v.double()
    eg = ((Gp[:n].double() - want).norm() / want.norm()).item()
    Gr = (A.double().t() @ A.double())
    ew = ((gram.double() - Gr).norm() / Gr.norm()).item()
    mask = (A > 0).double(); yhat = ((Yp - mu1) * is1).double()
    full = A.double() @ M.double() + v.double()
    m1 = (full * mask).sum(0); m2 = (full * mask * yhat).sum(0)
    got = sl.double().sum(0)
    e1 = ((got[0] - m1).norm() / m1.norm()).item(); e2 = ((got[1] - m2).norm() / m2.norm()).item()
    assert eg < 1e-5 and ew < 1e-5 and e1 < 1e-4 and e2 < 1e-4, (eg, ew, e1, e2)
    t_alg = timeit(alg)
    # today's pair for the same layer
    Y, W = rnd(P, Cout), rnd(Cout, Cin)
    cb, cd = rnd(Cout), rnd(Cout)
    dW = torch.zeros(Cout, Cin, device="cuda")
    t_pair = float("nan")
    if dll().prifit_gemm_stream_tn_supported(Cout, Cin, LL(P)) and dll().prifit_gemm_stream_supported(1, P, Cin, Cout):
        sl2 = torch.empty(dll().prifit_gemm_stream_slabs(P, Cout), 2, Cin, device="cuda")
        ws2 = torch.empty(dll().prifit_gemm_stream_tn_workspace(Cout, Cin, LL(P)), device="cuda")
        arg = torch.randint(0, pool_K, (P // pool_K, Cout), device="cuda", generator=g, dtype=torch.int32); T = rnd(P // pool_K, Cout)
        bias_dw = torch.mv(W.t(), cd)
        def pair():
            call("prifit_gemm_stream_tn_pool_f32", Cout, Cin, LL(P), ptr(Y), LL(Cout), ptr(Yp), LL(Cin), ptr(dW), LL(Cin), ptr(s1), ptr(t1), ptr(arg), ptr(T), ptr(cb), ptr(cd), pool_K, ptr(ws2), cur_stream())
            call("prifit_gemm_stream_dgrad_pool_f32", P, Cin, Cout, ptr(Y), LL(Cout), ptr(W), LL(Cin), ptr(Gp), LL(Cin), ptr(bias_dw), ptr(arg), ptr(T), ptr(cb), pool_K, ptr(Yp), LL(Cin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl2), cur_stream())
        t_pair = timeit(pair)
    t_fused = float("nan")
    if dll().prifit_pool_alg_fused_supported(LL(P), pool_K, Cout, Cin):
        G_ = P // pool_K
        arg2 = torch.randint(0, pool_K, (G_, Cout), device="cuda", generator=g, dtype=torch.int32)
        T2 = rnd(G_, Cout) * (torch.rand(G_, Cout, device="cuda", generator=g) > 0.4)      # ~40 % of the pooled gradients masked
        W2 = rnd(Cout, Cin)
        dWs = torch.empty(Cout, Cin, device="cuda"); asum = torch.empty(Cin, device="cuda")
        wsf = torch.empty(dll().prifit_pool_alg_fused_workspace(LL(P), Cout, Cin), device="cuda")
        def fused():
            call("prifit_pool_alg_fused_f32", LL(P), pool_K, Cout, Cin, ptr(Yp), LL(Cin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(M), LL(Cin),
                 ptr(v), ptr(arg2), ptr(T2), ptr(W2), ptr(Gp), LL(Cin), ptr(sl), ptr(gram), ptr(asum), ptr(dWs), ptr(wsf), cur_stream())
        t_fused = timeit(fused)
    t_sparse = float("nan")
    if dll().prifit_pool_alg_sparse_supported(P // pool_K, pool_K, Cout, Cin):
        G_ = P // pool_K
        arg3 = torch.randint(0, pool_K, (G_, Cout), device="cuda", generator=g, dtype=torch.int32)
        T3 = rnd(G_, Cout) * (torch.rand(G_, Cout, device="cuda", generator=g) > 0.4)
        W3 = rnd(Cout, Cin)
        dWs3 = torch.empty(Cout, Cin, device="cuda")
        sl3 = torch.empty(dll().prifit_pool_alg_sparse_slabs(G_), 2, Cin, device="cuda")
        ws3 = torch.empty(dll().prifit_pool_alg_sparse_workspace(G_, Cout, Cin), device="cuda")
        def sparse():
            call("prifit_pool_alg_sparse_f32", G_, pool_K, Cout, Cin, ptr(arg3), ptr(T3), ptr(W3), ptr(Yp), LL(Cin), ptr(s1), ptr(t1),
                 ptr(mu1), ptr(is1), ptr(Gp), LL(Cin), ptr(sl3), ptr(dWs3), ptr(ws3), cur_stream())
        t_sparse = timeit(sparse)
    print("   with the winners' rows inside the pass: %7.1f us;   winners' rows + channels as their own launches: %7.1f us" % (t_fused, t_sparse))
    print("[%8d rows, Cout %3d, Cin %3d] algebraic dense pass %7.1f us (%.0f GB/s of 8 P Cin B, %.1f TFLOP/s of its %.1f GFLOP)   today's pooled pair %7.1f us   errors %.1e %.1e %.1e %.1e"
          % (P, Cout, Cin, t_alg, 8.0 * P * Cin / t_alg / 1e3, 2.0 * P * Cin * Cin * 1.0 * (1 + (Cin // 32 + 1) / (2.0 * (Cin // 32))) / t_alg / 1e6,
             2.0 * P * Cin * Cin * (1 + (Cin // 32 + 1) / (2.0 * (Cin // 32))) / 1e9, t_pair, eg, ew, e1, e2), flush=True)
