"""GPU parity: MFMA GEMM, shared-MLP autograd, SA / FP modules and the full MSG network vs the oracle
(torch-CPU restatement) and the golden vectors captured from the reference.

Tolerances (fp32): GEMM 2e-5 relative to |A||B| row/col norms; module outputs 1e-4; module gradients
2e-4 (+ abs floor scaled by the largest gradient, see oracle/make_golden.py) at toy shapes; at the network's REAL
shapes (module_sa_real.npz) and for the whole network the gradient bars are 2 x the fp32 irreproducibility measured
between reference-fp32, oracle-fp32 and oracle-fp64 on the same inputs (sparse ReLU-mask / max-pool-winner flips):
5.6e-3 on whole-network gradient norms, 1.1e-2 / 1.2e-2 (relative L2) on the SA1 / SA2 parameter gradients."""
import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def nn_ops(hiplib):
    assert torch.cuda.is_available()
    from prifit_amd import nn_ops as m
    return m


def _rand(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).normal(size=shape).astype(np.float32))


@pytest.mark.parametrize("lay", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 50, 70), (1000, 196, 324), (64, 32, 6), (37, 129, 515),
                                   (2048, 2048, 128)])
def test_gemm_layouts(nn_ops, lay, M, N, K):
    A = _rand((M, K), 1)
    B = _rand((N, K), 2)
    ref = (A.double() @ B.double().T).float()
    Ad = (A if lay != 2 else A.T.contiguous()).cuda()
    Bd = (B if lay == 0 else B.T.contiguous()).cuda()
    C = torch.empty(M, N, device="cuda")
    nn_ops.gemm(lay, M, N, K, Ad, Ad.stride(0), Bd, Bd.stride(0), C, N)
    tol = 2e-5 * (A.norm(dim=1, keepdim=True) * B.norm(dim=1).unsqueeze(0))
    assert ((C.cpu() - ref).abs() <= tol + 1e-6).all()


def test_gemm_prologue_bias_stats_splitk_batched(nn_ops):
    M, N, K = 777, 96, 64
    A, W = _rand((M, K), 3), _rand((N, K), 4)
    sc, sh, bias = _rand((K,), 5), _rand((K,), 6), _rand((N,), 7)
    An = torch.relu(A * sc + sh)
    ref = (An.double() @ W.double().T).float() + bias
    Ad, Wd = A.cuda(), W.cuda()
    C = torch.empty(M, N, device="cuda")
    nslab = nn_ops.gemm_stats_slabs(M, N, K)
    slab = torch.zeros(nslab, 2, N, device="cuda")
    nn_ops.gemm(0, M, N, K, Ad, K, Wd, K, C, N, a_affine=(sc.cuda(), sh.cuda()), bias=bias.cuda(), stats=slab)
    torch.testing.assert_close(C.cpu(), ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(slab[:, 0].sum(0).cpu(), ref.sum(0), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(slab[:, 1].sum(0).cpu(), (ref * ref).sum(0), rtol=1e-4, atol=1e-2)
    # TN with prologue on B and split-K (the dW shape): dW = dY^T relu(bn(A))
    dY = _rand((M, N), 8)
    refW = (dY.double().T @ An.double()).float()
    dW = torch.zeros(N, K, device="cuda")
    nn_ops.gemm(2, N, K, M, dY.cuda(), N, Ad, K, dW, K, splitk=5, b_affine=(sc.cuda(), sh.cuda()))
    torch.testing.assert_close(dW.cpu(), refW, rtol=1e-4, atol=1e-3)
    # batched NT with the chord / mean-shift-kernel epilogues
    Bt, n, d = 3, 200, 128
    X = torch.nn.functional.normalize(_rand((Bt, n, d), 9), dim=-1)
    Xd = X.cuda()
    out = torch.empty(Bt, n, n, device="cuda")
    nn_ops.gemm(0, n, n, d, Xd, d, Xd, d, out, n, batch=Bt, sA=n * d, sB=n * d, sC=n * n, epi=1)
    torch.testing.assert_close(out.cpu(), 2 - 2 * X @ X.transpose(1, 2), rtol=1e-5, atol=2e-6)
    bw = torch.tensor([0.3, 0.5, 0.9])
    nn_ops.gemm(0, n, n, d, Xd, d, Xd, d, out, n, batch=Bt, sA=n * d, sB=n * d, sC=n * n, epi=2, epi_scalar=bw.cuda())
    refk = torch.exp(torch.clamp(-(2 - 2 * X @ X.transpose(1, 2)) / (bw.view(3, 1, 1) ** 2) / 2, -13, 75))
    torch.testing.assert_close(out.cpu(), refk, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("lay,M,N,K,aff", [(0, 40000, 96, 64, True), (0, 65536, 128, 96, True), (0, 33001, 64, 64, False),
                                           (0, 32768, 128, 128, True), (0, 50000, 64, 128, True),
                                           (1, 40000, 96, 128, False), (1, 33333, 64, 96, False), (1, 70000, 128, 64, False)])
def test_gemm_stream_tall_skinny(nn_ops, lay, M, N, K, aff):
    """Weights-stationary streaming kernel (csrc/gemm_stream.hip) on the shared-MLP shapes: forward (NT, BatchNorm+ReLU
    prologue, bias, column statistics) and dA (NN) against float64, ragged M; and against the tiled kernel."""
    if not nn_ops._STREAM:
        pytest.skip("PRIFIT_GEMM_STREAM=0")
    assert nn_ops._stream_ok(lay, M, N, K)
    A, W = _rand((M, K), 11), _rand((N, K), 12)
    sc, sh, bias = _rand((K,), 13), _rand((K,), 14), _rand((N,), 15)
    An = torch.relu(A * sc + sh) if aff else A
    ref = (An.double() @ W.double().T).float() + (bias if lay == 0 else 0)
    Ad = A.cuda()
    Wd = (W if lay == 0 else W.T.contiguous()).cuda()
    C = torch.empty(M, N, device="cuda")
    nslab = nn_ops.gemm_stats_slabs(M, N, K) if lay == 0 else 1
    slab = torch.full((nslab, 2, N), float("nan"), device="cuda") if lay == 0 else None
    kw = dict(a_affine=(sc.cuda(), sh.cuda())) if aff else {}
    if lay == 0:
        kw.update(bias=bias.cuda(), stats=slab)
    nn_ops.gemm(lay, M, N, K, Ad, K, Wd, Wd.stride(0), C, N, **kw)
    tol = 2e-5 * (An.norm(dim=1, keepdim=True) * W.norm(dim=1).unsqueeze(0))
    assert ((C.cpu() - ref).abs() <= tol + 1e-5).all()
    if lay == 0:
        torch.testing.assert_close(slab[:, 0].double().sum(0).cpu(), ref.double().sum(0), rtol=1e-4, atol=5e-2)
        torch.testing.assert_close(slab[:, 1].double().sum(0).cpu(), (ref.double() ** 2).sum(0), rtol=1e-4, atol=5e-2)
    old = nn_ops._STREAM
    nn_ops._STREAM = False
    try:
        C2 = torch.empty(M, N, device="cuda")
        kw2 = dict(kw)
        if lay == 0:
            kw2["stats"] = torch.empty(nn_ops.gemm_stats_slabs(M, N, K), 2, N, device="cuda")
        nn_ops.gemm(lay, M, N, K, Ad, K, Wd, Wd.stride(0), C2, N, **kw2)
    finally:
        nn_ops._STREAM = old
    assert torch.equal(C, C2)  # same k-ordered fp32 MFMA chain -> bit-identical products


@pytest.mark.parametrize("lay,M,N,K,aff", [(0, 70001, 256, 196, True), (0, 66000, 196, 128, True), (0, 65600, 256, 128, False),
                                           (1, 70001, 196, 256, False), (1, 65600, 128, 196, False)])
def test_gemm_persistent_tiles(nn_ops, lay, M, N, K, aff):
    """More output tiles than resident workgroups: the persistent 128 x 128 kernel (csrc/gemm.hip, gemm_pers_kernel) with
    ragged M, K = 196 / N = 196 (partial last k-tile, partial column tile), prologue, bias and column statistics, against
    float64."""
    A, W = _rand((M, K), 51), _rand((N, K), 52)
    sc, sh, bias = _rand((K,), 53), _rand((K,), 54), _rand((N,), 55)
    An = torch.relu(A * sc + sh) if aff else A
    ref = An.double() @ W.double().T + (bias.double() if lay == 0 else 0)
    Ad = A.cuda()
    Wd = (W if lay == 0 else W.T.contiguous()).cuda()
    C = torch.full((M, N), float("nan"), device="cuda")
    kw = dict(a_affine=(sc.cuda(), sh.cuda())) if aff else {}
    slab = None
    if lay == 0:
        slab = torch.full((nn_ops.gemm_stats_slabs(M, N, K), 2, N), float("nan"), device="cuda")
        kw.update(bias=bias.cuda(), stats=slab)
    nn_ops.gemm(lay, M, N, K, Ad, K, Wd, Wd.stride(0), C, N, **kw)
    tol = 2e-5 * (An.norm(dim=1, keepdim=True) * W.norm(dim=1).unsqueeze(0))
    assert ((C.cpu().double() - ref).abs() <= tol + 1e-5).all()
    if lay == 0:
        torch.testing.assert_close(slab[:, 0].double().sum(0).cpu(), ref.sum(0), rtol=1e-4, atol=5e-2)
        torch.testing.assert_close(slab[:, 1].double().sum(0).cpu(), (ref ** 2).sum(0), rtol=1e-4, atol=5e-2)


@pytest.mark.parametrize("Mo,No,P,aff", [(128, 96, 40008, True), (96, 64, 65536, True), (64, 64, 33000, False),
                                         (128, 64, 100000, True), (64, 32, 32768, True), (32, 32, 50000, False),
                                         (128, 128, 40000, True), (96, 96, 36000, False)])
def test_gemm_stream_weight_grad(nn_ops, Mo, No, P, aff):
    """LDS-free streaming dW kernel: dW = dY^T relu(bn(A)) over P rows against float64; accumulates into `out`."""
    if not nn_ops._STREAM:
        pytest.skip("PRIFIT_GEMM_STREAM=0")
    assert nn_ops.dll().prifit_gemm_stream_tn_supported(Mo, No, nn_ops._LL(P))
    dY, A = _rand((P, Mo), 21), _rand((P, No), 22)
    sc, sh = _rand((No,), 23), _rand((No,), 24)
    An = torch.relu(A * sc + sh) if aff else A
    ref = dY.double().T @ An.double()
    init = _rand((Mo, No), 25)
    out = init.clone().cuda()
    got = nn_ops._weight_grad(dY.cuda(), P, Mo, A.cuda(), No, (sc.cuda(), sh.cuda()) if aff else None, out=out)
    assert got is out
    err = (out.cpu().double() - init.double() - ref).norm() / ref.norm()
    assert err < 2e-6, err
    old = nn_ops._STREAM
    nn_ops._STREAM = False
    try:
        tiled = nn_ops._weight_grad(dY.cuda(), P, Mo, A.cuda(), No, (sc.cuda(), sh.cuda()) if aff else None)
    finally:
        nn_ops._STREAM = old
    assert ((tiled.cpu().double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("P,K,dims", [(65536, 64, ((64, 64), (64, 96), (96, 128))),     # streaming dA products, pooled fusion
                                      (131072, 128, ((64, 64), (64, 128))),                # pooled fusion, 128-sample groups, N = 64
                                      (65536, 32, ((64, 96), (96, 128))),                  # K = 32: pool_bwd_apply stays, pool candidates
                                      (98304, 96, ((64, 64), (64, 128))),                  # 3 candidate blocks per group
                                      (6144, 32, ((64, 196), (196, 256))),                 # tiled kernel, 128-row tiles
                                      (69632, 32, ((64, 196), (196, 256))),                # persistent tiled kernel (> 512 tiles)
                                      (3072, 0, ((516, 256), (256, 512), (512, 1024)))])   # tiled kernel, 64-row tiles
def test_shared_mlp_fused_bn_reduce_matches_separate_launches(nn_ops, P, K, dims):
    """The BatchNorm-backward column sums emitted by the dA epilogues (prifit_gemm_stream_dgrad_f32 /
    prifit_gemm_dgrad_bnred_f32) against the separate bn_relu_bwd_reduce launches, and the pooled last layer's dY formed
    in the streaming consumers (prifit_gemm_stream_dgrad_pool_f32 / _tn_pool_f32) against pool_bwd_apply, and a middle
    layer's dY formed in its consumers (prifit_gemm_stream_dgrad_bn_f32 / _tn_bn_f32) against bn_relu_bwd_apply: same
    gradients up to summation order."""
    x = _rand((P, dims[0][0]), 31).cuda()
    g = torch.Generator().manual_seed(32)
    tens = []
    for cin, cout in dims:
        tens += [(torch.randn(cout, cin, generator=g) * (2.0 / cin ** 0.5)).cuda().requires_grad_(True),
                 torch.zeros(cout, device="cuda", requires_grad=True),
                 (torch.rand(cout, generator=g) + 0.5).cuda().requires_grad_(True), (torch.randn(cout, generator=g) * 0.1).cuda().requires_grad_(True),
                 torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")]
    gout = _rand((P // K if K else P, dims[-1][1]), 33).cuda()
    res = {}
    for fuse in (True, False):
        old = nn_ops._FUSE_RED, nn_ops._FUSE_POOL, nn_ops._FUSE_POOL_FWD, nn_ops._FUSE_BN_APPLY
        nn_ops._FUSE_RED = nn_ops._FUSE_POOL = nn_ops._FUSE_POOL_FWD = nn_ops._FUSE_BN_APPLY = fuse
        try:
            xi = x.clone().requires_grad_(True)
            cfg = {"pool_K": K, "training": True, "eps": 1e-5, "momentum": [0.1] * len(dims)}
            out = nn_ops.SharedMLPFn.apply(xi, cfg, *[t.clone() if not t.requires_grad else t for t in tens])
            grads = torch.autograd.grad(out, [xi] + [t for t in tens if t.requires_grad], gout, allow_unused=True)
            res[fuse] = [out.detach()] + [None if gg is None else gg.detach().clone() for gg in grads]
        finally:
            nn_ops._FUSE_RED, nn_ops._FUSE_POOL, nn_ops._FUSE_POOL_FWD, nn_ops._FUSE_BN_APPLY = old
    for a, b in zip(res[True], res[False]):
        if b is None:
            assert a is None
            continue
        assert (a - b).norm() <= 5e-5 * b.norm() + 1e-7, ((a - b).norm().item(), b.norm().item())


@pytest.mark.parametrize("P,K,dims", [(98304, 128, ((64, 64), (64, 96), (96, 128))), (65536, 64, ((64, 64), (64, 64), (64, 128))),
                                      (32768, 64, ((64, 128), (128, 128), (128, 128)))])
def test_shared_mlp_fused_da_dw_matches_separate_kernels(nn_ops, P, K, dims):
    """SharedMLPFn with the one-pass dA + dW kernel (default) against the separate streaming dA / dW kernels: same forward,
    every gradient to fp32 rounding."""
    x = _rand((P, dims[0][0]), 31).cuda()
    g = torch.Generator().manual_seed(32)
    tens = []
    for cin, cout in dims:
        tens += [(torch.randn(cout, cin, generator=g) * (2.0 / cin ** 0.5)).cuda().requires_grad_(True),
                 torch.zeros(cout, device="cuda", requires_grad=True),
                 (torch.rand(cout, generator=g) + 0.5).cuda().requires_grad_(True), (torch.randn(cout, generator=g) * 0.1).cuda().requires_grad_(True),
                 torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")]
    gout = _rand((P // K, dims[-1][1]), 33).cuda()
    res = {}
    for fuse in (True, False):
        old = nn_ops._FUSE_BWD
        nn_ops._FUSE_BWD = "1" if fuse else "0"
        try:
            xi = x.clone().requires_grad_(True)
            cfg = {"pool_K": K, "training": True, "eps": 1e-5, "momentum": [0.1] * len(dims)}
            out = nn_ops.SharedMLPFn.apply(xi, cfg, *[t.clone() if not t.requires_grad else t for t in tens])
            grads = torch.autograd.grad(out, [xi] + [t for t in tens if t.requires_grad], gout, allow_unused=True)
            res[fuse] = [out.detach()] + [None if gg is None else gg.detach().clone() for gg in grads]
        finally:
            nn_ops._FUSE_BWD = old
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        if b is None:
            assert a is None
            continue
        assert (a - b).norm() <= 2e-5 * b.norm() + 1e-7, ((a - b).norm().item(), b.norm().item())


@pytest.mark.parametrize("P,Cout,Kin", [(40008, 96, 64), (65536, 64, 64), (33000, 128, 128), (50000, 128, 96)])
def test_bn_apply_on_load_matches_apply_pass(nn_ops, P, Cout, Kin):
    """prifit_gemm_stream_tn_bn_f32 / prifit_gemm_stream_dgrad_bn_f32 (a middle layer's dY formed from G and Y inside the
    dW / dA kernels) against prifit_bn_relu_bwd_apply followed by the plain streaming kernels: the SAME numbers (the
    operand is the apply kernel's expression term by term, the products run in the same order); ragged P."""
    from prifit_amd.nn_ops import call, ptr, cur_stream, _LL, _F, dll
    G, Y, A = _rand((P, Cout), 61).cuda(), _rand((P, Cout), 62).cuda(), _rand((P, Kin), 63).cuda()
    W = _rand((Cout, Kin), 64).cuda()
    s, t, ca, cb, cd = [_rand((Cout,), 65 + i).cuda() for i in range(5)]
    s1, t1, mu1, is1 = [_rand((Kin,), 71 + i).cuda() for i in range(4)]
    dY = torch.empty(P, Cout, device="cuda")
    call("prifit_bn_relu_bwd_apply", ptr(G), _LL(Cout), ptr(Y), _LL(Cout), ptr(s), ptr(t), ptr(ca), ptr(cb), ptr(cd), P, Cout, 0,
         _F(0.0), ptr(dY), _LL(Cout), cur_stream())
    # dW
    ws = torch.empty(dll().prifit_gemm_stream_tn_workspace(Cout, Kin, _LL(P)), device="cuda")
    dW_ref, dW = torch.zeros(Cout, Kin, device="cuda"), torch.zeros(Cout, Kin, device="cuda")
    call("prifit_gemm_stream_tn_f32", Cout, Kin, _LL(P), ptr(dY), _LL(Cout), ptr(A), _LL(Kin), ptr(dW_ref), _LL(Kin), ptr(s1), ptr(t1),
         ptr(ws), cur_stream())
    call("prifit_gemm_stream_tn_bn_f32", Cout, Kin, _LL(P), ptr(G), ptr(Y), _LL(Cout), ptr(A), _LL(Kin), ptr(dW), _LL(Kin), ptr(s1),
         ptr(t1), ptr(s), ptr(t), ptr(ca), ptr(cb), ptr(cd), ptr(ws), cur_stream())
    assert (dW - dW_ref).norm() <= 2e-6 * dW_ref.norm()      # (slab sums may be added in a different order)
    # dA with the BatchNorm-backward partials of the previous layer
    ns = dll().prifit_gemm_stream_slabs(P, Cout)
    Gp_ref, Gp = torch.empty(P, Kin, device="cuda"), torch.full((P, Kin), float("nan"), device="cuda")
    sl_ref, sl = torch.empty(ns, 2, Kin, device="cuda"), torch.full((ns, 2, Kin), float("nan"), device="cuda")
    call("prifit_gemm_stream_dgrad_f32", P, Kin, Cout, ptr(dY), _LL(Cout), ptr(W), _LL(Kin), ptr(Gp_ref), _LL(Kin), ptr(A), _LL(Kin),
         ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl_ref), None, cur_stream())
    call("prifit_gemm_stream_dgrad_bn_f32", P, Kin, Cout, ptr(G), ptr(Y), _LL(Cout), ptr(W), _LL(Kin), ptr(Gp), _LL(Kin), ptr(s), ptr(t),
         ptr(ca), ptr(cb), ptr(cd), ptr(A), _LL(Kin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl), None, cur_stream())
    assert torch.equal(Gp, Gp_ref)
    torch.testing.assert_close(sl.double().sum(0), sl_ref.double().sum(0), rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("P,Cout,Kin,pool_K", [(40008, 96, 64, 0), (65536, 64, 64, 0), (33000, 128, 128, 0), (50000, 128, 96, 0),
                                              (36936, 128, 64, 0), (49152, 128, 96, 128), (65536, 128, 64, 64), (40960, 64, 64, 64),
                                              (36864, 128, 128, 192)])
def test_fused_da_dw_kernel_matches_separate_streaming_kernels(nn_ops, P, Cout, Kin, pool_K):
    """prifit_gemm_stream_bwd_f32 (dA, the BatchNorm-backward partials of the layer below and dW from ONE pass over the rows)
    against the separate streaming kernels on the same operands: Gp, dW and the (m1, m2) sums to fp32 rounding (the
    products are summed in another order); middle-layer and max-pooled forms, ragged P, every supported (Cout, Cin)."""
    from prifit_amd.nn_ops import call, ptr, cur_stream, _LL, _F, dll
    assert dll().prifit_gemm_stream_bwd_supported(_LL(P), Cout, Kin, pool_K)
    Y, A = _rand((P, Cout), 62).cuda(), _rand((P, Kin), 63).cuda()
    W = _rand((Cout, Kin), 64).cuda()
    s, t, ca, cb, cd = [_rand((Cout,), 65 + i).cuda() for i in range(5)]
    s1, t1, mu1, is1 = [_rand((Kin,), 71 + i).cuda() for i in range(4)]
    ns_ref = dll().prifit_gemm_stream_slabs(P, Cout)
    Gp_ref = torch.empty(P, Kin, device="cuda")
    sl_ref = torch.empty(ns_ref, 2, Kin, device="cuda")
    dW_ref = torch.zeros(Cout, Kin, device="cuda")
    ws = torch.empty(dll().prifit_gemm_stream_tn_workspace(Cout, Kin, _LL(P)), device="cuda")
    if pool_K:
        Gn = P // pool_K
        arg = torch.randint(0, pool_K, (Gn, Cout), generator=torch.Generator().manual_seed(5), dtype=torch.int32).cuda()
        T = _rand((Gn, Cout), 81).cuda()
        bias_dw = torch.mv(W.t(), cd)
        call("prifit_gemm_stream_tn_pool_f32", Cout, Kin, _LL(P), ptr(Y), _LL(Cout), ptr(A), _LL(Kin), ptr(dW_ref), _LL(Kin), ptr(s1),
             ptr(t1), ptr(arg), ptr(T), ptr(cb), ptr(cd), pool_K, ptr(ws), cur_stream())
        call("prifit_gemm_stream_dgrad_pool_f32", P, Kin, Cout, ptr(Y), _LL(Cout), ptr(W), _LL(Kin), ptr(Gp_ref), _LL(Kin), ptr(bias_dw),
             ptr(arg), ptr(T), ptr(cb), pool_K, ptr(A), _LL(Kin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl_ref), None, cur_stream())
        G = None
    else:
        G = _rand((P, Cout), 61).cuda()
        arg = T = None
        call("prifit_gemm_stream_tn_bn_f32", Cout, Kin, _LL(P), ptr(G), ptr(Y), _LL(Cout), ptr(A), _LL(Kin), ptr(dW_ref), _LL(Kin), ptr(s1),
             ptr(t1), ptr(s), ptr(t), ptr(ca), ptr(cb), ptr(cd), ptr(ws), cur_stream())
        call("prifit_gemm_stream_dgrad_bn_f32", P, Kin, Cout, ptr(G), ptr(Y), _LL(Cout), ptr(W), _LL(Kin), ptr(Gp_ref), _LL(Kin), ptr(s), ptr(t),
             ptr(ca), ptr(cb), ptr(cd), ptr(A), _LL(Kin), ptr(s1), ptr(t1), ptr(mu1), ptr(is1), ptr(sl_ref), None, cur_stream())
    ns = dll().prifit_gemm_stream_bwd_slabs(_LL(P), Cout, Kin)
    Gp = torch.full((P, Kin), float("nan"), device="cuda")
    sl = torch.full((ns, 2, Kin), float("nan"), device="cuda")
    dW = torch.full((Cout, Kin), float("nan"), device="cuda")
    ws2 = torch.empty(dll().prifit_gemm_stream_bwd_workspace(_LL(P), Cout, Kin), device="cuda")
    call("prifit_gemm_stream_bwd_f32", _LL(P), Cout, Kin, ptr(G), ptr(Y), ptr(None if pool_K else s), ptr(None if pool_K else t),
         ptr(None if pool_K else ca), ptr(cb), ptr(cd), ptr(arg), ptr(T), pool_K, ptr(W), _LL(Kin), ptr(A), _LL(Kin), ptr(s1), ptr(t1),
         ptr(mu1), ptr(is1), ptr(Gp), _LL(Kin), ptr(sl), ptr(dW), _LL(Kin), ptr(ws2), None, cur_stream())
    assert torch.isfinite(Gp).all() and torch.isfinite(dW).all() and torch.isfinite(sl).all()
    assert (Gp - Gp_ref).abs().max() <= 2e-5 * Gp_ref.abs().max()
    assert (dW - dW_ref).norm() <= 2e-6 * dW_ref.norm()
    torch.testing.assert_close(sl.double().sum(0), sl_ref.double().sum(0), rtol=2e-5, atol=2e-5 * sl_ref.double().sum(0).abs().max().item())


@pytest.mark.parametrize("P,K,N,Kin", [(65536, 64, 128, 64), (49152, 96, 64, 96), (32768, 32, 96, 128)])
def test_pool_candidates_match_pool_fwd(nn_ops, P, K, N, Kin):
    """prifit_gemm_stream_pool_f32 + prifit_pool_from_candidates against prifit_pool_fwd on the stored Y: identical pooled
    values; identical winners wherever the pooled activation is positive (rows duplicated inside groups -> ties -> first)."""
    from prifit_amd.nn_ops import call, ptr, cur_stream, _LL, _F
    A, W = _rand((P, Kin), 41), _rand((N, Kin), 42)
    A.view(P // K, K, Kin)[:, K // 2:] = A.view(P // K, K, Kin)[:, :1]          # padding-like duplicates of sample 0
    sc, sh, bias = _rand((Kin,), 43), _rand((Kin,), 44), _rand((N,), 45)
    s2, t2 = _rand((N,), 46), _rand((N,), 47)                                   # both signs of the BatchNorm scale
    s2[3] = 0.0
    Ad, Wd = A.cuda(), W.cuda()
    Y = torch.empty(P, N, device="cuda")
    cand = torch.empty(P // 32, 4, N, device="cuda")
    slab = torch.empty(nn_ops.gemm_stats_slabs(P, N, Kin), 2, N, device="cuda")
    call("prifit_gemm_stream_pool_f32", P, N, Kin, ptr(Ad), _LL(Kin), ptr(Wd), _LL(Kin), ptr(Y), _LL(N), ptr(sc.cuda()),
         ptr(sh.cuda()), ptr(bias.cuda()), ptr(slab), ptr(cand), None, cur_stream())
    G = P // K
    s2d, t2d = s2.cuda(), t2.cuda()
    out1, arg1 = torch.empty(G, N, device="cuda"), torch.empty(G, N, dtype=torch.int32, device="cuda")
    out2, arg2 = torch.empty(G, N, device="cuda"), torch.empty(G, N, dtype=torch.int32, device="cuda")
    ystar = torch.empty(G, N, device="cuda")
    call("prifit_pool_from_candidates", ptr(cand), ptr(s2d), ptr(t2d), G, K, N, 0, _F(0.0), ptr(out1), _LL(N), ptr(arg1), ptr(ystar), cur_stream())
    call("prifit_pool_fwd", ptr(Y), _LL(N), ptr(s2d), ptr(t2d), G, K, N, 0, _F(0.0), ptr(out2), _LL(N), ptr(arg2), cur_stream())
    assert torch.equal(out1, out2)
    live = out2 > 0
    assert torch.equal(arg1[live], arg2[live])
    assert live.float().mean() > 0.2
    # ystar = the stored product at the winner (where the scale is not zero: there the first row wins by convention)
    want = torch.gather(Y.view(G, K, N), 1, arg1.long().unsqueeze(1)).squeeze(1)
    nz = (s2d != 0).view(1, N).expand(G, N)
    assert torch.equal(ystar[nz], want[nz])


def _run_pair(my, orc_mod, args_gpu, args_cpu, gout, n_out=1, pick=lambda o: o):
    """Run the HIP module and the oracle module with identical parameters; return outputs+grads."""
    my.load_state_dict(orc_mod.state_dict())
    my.cuda().train()
    orc_mod.train()
    o_ref = pick(orc_mod(*args_cpu))
    (o_ref * gout).sum().backward()
    o_my = pick(my(*args_gpu))
    (o_my * gout.cuda()).sum().backward()
    return o_my, o_ref


def _check_param_grads(my, ref, rtol=2e-4):
    rg = {k: p.grad for k, p in ref.named_parameters()}
    gmax = max(v.abs().max().item() for v in rg.values())
    for k, p in my.named_parameters():
        torch.testing.assert_close(p.grad.cpu(), rg[k], rtol=rtol, atol=2e-5 * gmax, msg=lambda m: k + ": " + m)
    for (k, b1), (_, b2) in zip(my.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(b1.cpu().float(), b2.float(), rtol=1e-4, atol=1e-5, msg=lambda m: k + ": " + m)


def test_pack_cols_forward_and_backward(hiplib):
    """PackColsFn (prifit_pack_cols / prifit_unpack_cols): a column map with a permutation, zero padding and a source column that
    feeds two output columns, against plain indexing and its autograd."""
    from prifit_amd.models.pointnet_util import PackColsFn
    gen = torch.Generator().manual_seed(3)
    w = torch.randn(37, 9, generator=gen)
    cols = (3, 4, 5, 0, 1, 2, -1, 8, 8, -1, 6, -1)          # column 7 unused, column 8 twice, three pad columns
    go = torch.randn(37, len(cols), generator=gen)
    wr = w.clone().requires_grad_(True)
    idx = torch.tensor([max(c, 0) for c in cols])
    mask = torch.tensor([1.0 if c >= 0 else 0.0 for c in cols])
    ref = wr[:, idx] * mask
    (ref * go).sum().backward()
    wd = w.cuda().requires_grad_(True)
    out = PackColsFn.apply(wd, cols)
    (out * go.cuda()).sum().backward()
    assert torch.equal(out.detach().cpu(), ref.detach())
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-6, atol=1e-6)
    assert torch.equal(wd.grad[:, 7].cpu(), torch.zeros(37))


def test_packed_weights_in_one_launch_give_the_same_step(hiplib, monkeypatch):
    """Round 6: the column-packed first-layer weights of the MSG network come from ONE launch at the top of the forward
    (PackAllFn; their gradients from one at the end of the backward) from the second forward on -- the first one discovers
    the sites.  Copies and fixed-order sums: the same loss bit for bit, the same gradients (to the rounding noise of the atomics in the
    weight-gradient reductions behind them), and the plan holds all the sites the forward asks for."""
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.models import pointnet_util as PU
    B, N = 2, 1024
    xyz = _t(synth.cloud("surface", B, N, 5)).transpose(1, 2).contiguous().cuda()
    cls = torch.zeros(B, 1, 16, device="cuda")
    cls[:, 0, 2] = 1.0
    target = _t(synth.labels(B, N, 50, 5)).cuda()
    s1, s2 = _t(synth.fps_start(B, N, 5)).cuda(), _t(synth.fps_start(B, 512, 6)).cuda()
    res = []
    for all_at_once in (True, False):
        monkeypatch.setattr(PU, "_PACK_ALL", all_at_once)
        torch.manual_seed(3)
        net = M.get_model(50)
        synth.xavier_like_trainer(net)
        net.cuda().train()
        net.drop1.eval()
        for step in range(2):                         # step 0 discovers the sites, step 1 runs from the plan
            for p in net.parameters():
                p.grad = None
            seg = net(xyz, cls, fps_start=(s1, s2))[0]
            loss = M.get_loss()(seg.contiguous().view(-1, 50), target.view(-1), None)
            loss.backward()
        plan = PU._plans[net]
        assert len(plan.sites) >= 6 and plan.results is None
        packed = {k for k, p in net.named_parameters() if any(p is q for q, _ in plan.sites)}
        res.append((loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, packed))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and res[0][2] == res[1][2] and len(res[0][2]) >= 6
    for k in res[0][1]:
        # (the weight gradients come from split-K / slab reductions with atomics: equal to rounding, run to run, with either form)
        torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=1e-4, atol=1e-7, msg=k)


def test_sa_msg_module(hiplib, golden):
    from prifit_amd.models import pointnet_util as pu
    g = golden("module_sa_msg")
    B, N, seed = 2, 512, int(g["seed"])
    xyz = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    feat = _t(synth.features(B, N, 16, seed)).transpose(1, 2).contiguous()
    start = _t(g["start"])
    torch.manual_seed(11)
    ref = orc.OracleSetAbstractionMsg(64, [0.2, 0.4], [8, 16], 16, [[16, 32], [16, 24, 32]])
    synth.perturb_bn(ref, 3)
    my = pu.PointNetSetAbstractionMsg(64, [0.2, 0.4], [8, 16], 16, [[16, 32], [16, 24, 32]])
    gout = _t(synth.features(B, 64, 64, seed + 1)).transpose(1, 2)
    f_cpu = feat.clone().requires_grad_(True)
    f_gpu = feat.cuda().requires_grad_(True)
    o_my, o_ref = _run_pair(my, ref, (xyz.cuda(), f_gpu, start.cuda()), (xyz, f_cpu, start), gout, pick=lambda o: o[1])
    torch.testing.assert_close(o_my.cpu(), _t(g["out"]), rtol=1e-4, atol=1e-4)   # vs the reference itself
    torch.testing.assert_close(o_my.cpu(), o_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f_gpu.grad.cpu(), _t(g["dfeat"]), rtol=2e-4, atol=1e-5)
    _check_param_grads(my, ref)
    torch.testing.assert_close(my.bn_blocks[0][0].running_var.cpu(), _t(g["running_var_00"]), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("i", [0, 1])
def test_sa_levels_at_real_shapes(hiplib, golden, i):
    """SA1 (B=4 x 2048 -> 512 centres, 3 scales) and SA2 (B=4 x 512 x 320 ch -> 128 centres, 2 scales) of the MSG network
    with the DEFAULT switches: fused grouping front end (direct mode / linearity), streaming GEMMs, fused BatchNorm /
    max-pool epilogues -- against the reference's outputs and every parameter gradient (module_sa_real.npz).  Output
    1e-4; gradients in relative L2 within the fixture's measured fp32-irreproducibility bound."""
    import sa_real_common as C
    from prifit_amd.models import pointnet_util as pu
    g = golden("module_sa_real")
    xyz, feat, start, gout = C.inputs(g, i)
    ref = C.seeded_module(orc.OracleSetAbstractionMsg, i)
    my = pu.PointNetSetAbstractionMsg(*C.CASES[i][3])
    my.load_state_dict(ref.state_dict())
    my.cuda().train()
    # SA1 in the network sees l0_points = xyz, which needs no gradient (models/pointnet2_part_seg_msg.py:69-75): that is
    # the path the benchmark runs (direct mode); SA2's input features do need one
    f = feat.cuda().requires_grad_(i == 1)
    nx, out = my(xyz.cuda(), f, start.cuda())
    (out * gout.cuda()).sum().backward()
    last = my.bn_blocks[-1][-1]
    rep = C.check(g, i, nx.cpu(), out.detach().cpu(), f.grad.cpu() if i == 1 else None,
                  {k: p.grad.cpu() for k, p in my.named_parameters()}, (last.running_mean.cpu(), last.running_var.cpu()))
    print(C.CASES[i][0], "worst relative L2 gradient deviation %.2e" % max(rep.values()))


def test_sa_group_all_and_ssg(hiplib, golden):
    from prifit_amd.models import pointnet_util as pu
    B, N, seed = 2, 512, 5
    xyz = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    feat = _t(synth.features(B, N, 16, seed)).transpose(1, 2).contiguous()
    start = _t(synth.fps_start(B, N, seed))
    for name, ctor, bnseed, s, gseed, ncol in (("module_sa_all", (None, None, None, 19, [32, 64], True), 4, 12, 2, 1),
                                                ("module_sa_ssg", (64, 0.3, 16, 19, [32, 64], False), 5, 13, 3, 64)):
        g = golden(name)
        torch.manual_seed(s)
        ref = orc.OracleSetAbstraction(*ctor)
        synth.perturb_bn(ref, bnseed)
        my = pu.PointNetSetAbstraction(*ctor)
        gout = _t(synth.features(B, ncol, 64, seed + gseed)).transpose(1, 2)
        f_cpu = feat.clone().requires_grad_(True)
        f_gpu = feat.cuda().requires_grad_(True)
        o_my, o_ref = _run_pair(my, ref, (xyz.cuda(), f_gpu, start.cuda()), (xyz, f_cpu, start), gout,
                                pick=lambda o: o[1])
        torch.testing.assert_close(o_my.cpu(), _t(g["out"]), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(f_gpu.grad.cpu(), _t(g["dfeat"]), rtol=2e-4, atol=1e-5)
        _check_param_grads(my, ref)


def test_fp_module(hiplib, golden):
    from prifit_amd.models import pointnet_util as pu
    B, N, S, seed = 2, 512, 64, 5
    xyz = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    xyz2 = xyz[:, :, :S].contiguous()
    p1 = _t(synth.features(B, N, 8, seed + 4)).transpose(1, 2).contiguous()
    p2 = _t(synth.features(B, S, 24, seed + 5)).transpose(1, 2).contiguous()
    gout = _t(synth.features(B, N, 16, seed + 6)).transpose(1, 2)
    for name, x2, q2 in (("module_fp", xyz2, p2), ("module_fp_s1", xyz2[:, :, :1].contiguous(), p2[:, :, :1].contiguous())):
        g = golden(name)
        torch.manual_seed(14)
        ref = orc.OracleFeaturePropagation(32, [32, 16])
        synth.perturb_bn(ref, 6)
        my = pu.PointNetFeaturePropagation(32, [32, 16])
        a1, a2 = p1.clone().requires_grad_(True), q2.clone().requires_grad_(True)
        b1, b2 = p1.cuda().requires_grad_(True), q2.cuda().requires_grad_(True)
        o_my, o_ref = _run_pair(my, ref, (xyz.cuda(), x2.cuda(), b1, b2), (xyz, x2, a1, a2), gout)
        torch.testing.assert_close(o_my.cpu(), _t(g["out"]), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(b2.grad.cpu(), _t(g["dpoints2"]), rtol=2e-4, atol=1e-5)
        torch.testing.assert_close(b1.grad.cpu(), a1.grad, rtol=2e-4, atol=1e-5)
        if name == "module_fp":
            _check_param_grads(my, ref)


def test_eval_mode_matches_oracle(hiplib):
    from prifit_amd.models import pointnet_util as pu
    B, N = 2, 256
    xyz = _t(synth.cloud("cube", B, N, 3)).transpose(1, 2).contiguous()
    feat = _t(synth.features(B, N, 8, 3)).transpose(1, 2).contiguous()
    start = _t(synth.fps_start(B, N, 3))
    torch.manual_seed(5)
    ref = orc.OracleSetAbstractionMsg(32, [0.3, 0.6], [8, 16], 8, [[16, 16], [16, 32]])
    synth.perturb_bn(ref, 9)
    my = pu.PointNetSetAbstractionMsg(32, [0.3, 0.6], [8, 16], 8, [[16, 16], [16, 32]])
    my.load_state_dict(ref.state_dict())
    my.cuda().eval()
    ref.eval()
    f_cpu = feat.clone().requires_grad_(True)
    f_gpu = feat.cuda().requires_grad_(True)
    o_ref = ref(xyz, f_cpu, start)[1]
    o_my = my(xyz.cuda(), f_gpu, start.cuda())[1]
    torch.testing.assert_close(o_my.cpu(), o_ref, rtol=1e-4, atol=1e-5)
    o_ref.sum().backward()
    o_my.sum().backward()
    torch.testing.assert_close(f_gpu.grad.cpu(), f_cpu.grad, rtol=2e-4, atol=1e-5)
    rg = {k: p.grad for k, p in ref.named_parameters()}
    for k, p in my.named_parameters():
        torch.testing.assert_close(p.grad.cpu(), rg[k], rtol=2e-4, atol=1e-4, msg=lambda m: k + ": " + m)


def test_full_model_supervised_step(hiplib, golden):
    """train_partseg_shapenet.py:382-399 on B=2 x 2048: loss, log-probs, feat and gradients."""
    from prifit_amd.models import pointnet2_part_seg_msg as M
    g = golden("model_msg_sup")
    B, N, seed = 2, 2048, int(g["seed"])
    torch.manual_seed(21)
    net = M.get_model(50)
    synth.xavier_like_trainer(net)
    synth.perturb_bn(net, 8)
    net.cuda().train()
    net.drop1.eval()
    xyz = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous().cuda()
    cls = torch.zeros(B, 1, 16, device="cuda")
    cls[:, 0, 3] = 1.0
    target = _t(synth.labels(B, N, 50, seed)).cuda()
    seg, (l1, l2, l3), feat, tl, cl = net(xyz, cls, fps_start=(_t(g["s1"]).cuda(), _t(g["s2"]).cuda()))
    assert seg.shape == (B, N, 50) and feat.shape == (B, 128, N) and l1.shape == (B, 128, 512) and l3.shape == (B, 1024, 1)
    loss = M.get_loss()(seg.contiguous().view(-1, 50), target.view(-1), None)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), _t(g["loss"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(seg[:, :64].detach().cpu(), _t(g["seg_head"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(feat[:, :, :64].detach().cpu(), _t(g["feat_head"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(l3.detach().cpu(), _t(g["l3"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(net.conv2.weight.grad.cpu(), _t(g["g_conv2_weight"]), rtol=1e-3, atol=1e-6)
    # bars = 2 x the fp32 irreproducibility MEASURED between reference-fp32, oracle-fp32 and oracle-fp64 on this very
    # step (oracle/make_golden.py:golden_model -> model_msg_sup_noise.npz: norms 2.8e-3, vectors 9.3e-3)
    noise = golden("model_msg_sup_noise")
    norm_tol, vec_tol = float(noise["norm_tol"]), float(noise["vec_tol"])
    assert norm_tol <= 1e-2 and vec_tol <= 2e-2
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    worst = 0.0
    for k, p in net.named_parameters():
        if k in norms and norms[k] > 0 and not (k.endswith(".bias") and "conv" in k and k != "conv2.bias"):
            assert p.grad is not None, k
            worst = max(worst, abs(p.grad.norm().item() - norms[k]) / norms[k])
            assert abs(p.grad.norm().item() - norms[k]) <= norm_tol * norms[k], (k, p.grad.norm().item(), norms[k])
    for k, ref in (("sa1.conv_blocks.0.0.weight", g["g_sa1_first"]), ("fp1.mlp_convs.1.weight", g["g_fp1_last"])):
        got = dict(net.named_parameters())[k].grad.cpu()
        rel = (got - _t(ref)).norm() / _t(ref).norm()
        assert rel < vec_tol, (k, rel)
    print("full model: worst gradient-norm deviation %.2e (bar %.2e)" % (worst, norm_tol))


def test_normal_channel_and_l2_norm_variants(hiplib):
    """models/pointnet2_part_seg_msg.py:12-25,69-75: normal_channel=True (6 input channels: sa1 sees [xyz+normals | rel],
    fp1's skip is 16+3+6) forward + backward against the oracle with the same parameters; l2_norm=True normalises the
    embedding once more before the loss, which normalises anyway -- the loss must not move."""
    from prifit_amd.models import pointnet2_part_seg_msg as M
    B, N = 2, 1024
    torch.manual_seed(31)
    ref = orc.OracleMSGPartSeg(50, normal_channel=True)
    synth.xavier_like_trainer(ref)
    synth.perturb_bn(ref, 6)
    net = M.get_model(50, normal_channel=True)
    net.load_state_dict(ref.state_dict())
    net.cuda().train()
    ref.train()
    net.drop1.eval()
    ref.drop1.eval()
    pts = _t(synth.cloud("surface", B, N, 33))
    nrm = torch.nn.functional.normalize(_t(synth.features(B, N, 3, 34)), dim=2)
    x6 = torch.cat([pts, nrm], 2).transpose(1, 2).contiguous()                 # [B,6,N]
    cls = torch.zeros(B, 1, 16)
    target = _t(synth.labels(B, N, 50, 35))
    s = (_t(synth.fps_start(B, N, 36)), _t(synth.fps_start(B, 512, 37)))
    seg_r = ref(x6, cls, fps_start=s)[0]
    loss_r = orc.seg_loss(seg_r.reshape(-1, 50), target.view(-1))
    loss_r.backward()
    out = net(x6.cuda(), cls.cuda(), fps_start=(s[0].cuda(), s[1].cuda()))
    loss = M.get_loss()(out[0].reshape(-1, 50), target.cuda().view(-1), None)
    loss.backward()
    assert abs(loss.item() - loss_r.item()) < 1e-5 * abs(loss_r.item())
    torch.testing.assert_close(out[0].detach().cpu(), seg_r.detach(), rtol=1e-3, atol=2e-4)
    for k, p in ref.named_parameters():
        if p.grad is None or (k.endswith(".bias") and "conv" in k and k != "conv2.bias"):
            continue
        g = dict(net.named_parameters())[k].grad.cpu()
        assert abs(g.norm().item() - p.grad.norm().item()) <= 1e-2 * p.grad.norm().item(), k
    assert net.sa1.conv_blocks[0][0].weight.shape[1] == 9 and net.fp1.mlp_convs[0].weight.shape[1] == 153
    # l2_norm=True with the convex loss
    from tests_helpers import fit_inputs
    _, cham, _ = fit_inputs(2, 1024, 128, 8)
    subset = cham[:, :1024]
    x = subset.transpose(1, 2).contiguous().cuda()
    losses = []
    for l2 in (False, True):
        torch.manual_seed(32)
        m = M.get_model(50, l2_norm=l2).cuda().train()
        R = _t(synth.uniform01((3, 3), 3)).cuda()
        o = m(x, torch.zeros(2, 1, 16, device="cuda"), chamfer_points=cham.transpose(1, 2).contiguous().cuda(),
              include_convex_loss=True, quantile=0.05, msc_iterations=5, max_num_clusters=25,
              fps_start=(s[0][:2].cuda(), s[1][:2].cuda()), fit_inputs=dict(rand_table=R))
        o[3].mean().backward()
        assert torch.isfinite(m.extra_conv_emb.weight.grad).all()
        losses.append(o[3].item())
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0]) + 1e-7, losses


def test_ssg_model_config1(hiplib, golden):
    """BASELINE.json configs[0]: PointNet++-SSG part-seg on 4 x 1024 clouds vs the reference's outputs
    (same seeded init: the parameter containers are constructed in the reference's order)."""
    from prifit_amd.models import pointnet2_part_seg_ssg as S
    g = golden("model_ssg")
    torch.manual_seed(22)
    net = S.get_model(50)
    assert sum(p.numel() for p in net.parameters()) == 1411250  # SURVEY.md 8a'' probe
    net.cuda().train()
    net.drop1.eval()
    xyz = _t(synth.cloud("cube", 4, 1024, int(g["seed"]))).transpose(1, 2).contiguous().cuda()
    seg, l3 = net(xyz, torch.zeros(4, 1, 16, device="cuda"), fps_start=(_t(g["s1"]).cuda(), _t(g["s2"]).cuda()))
    assert seg.shape == (4, 1024, 50) and l3.shape == (4, 1024, 1)
    torch.testing.assert_close(l3.detach().cpu(), _t(g["l3"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(seg.detach().sum(dim=1).cpu(), _t(g["seg_sum"]), rtol=1e-3, atol=5e-2)
    S.get_loss()(seg.reshape(-1, 50), torch.zeros(4096, dtype=torch.long, device="cuda")).backward()
    assert torch.isfinite(net.sa1.mlp_convs[0].weight.grad).all()


def test_model_variants(hiplib, golden):
    """cls MSG / cls SSG / sem-seg vs the reference's outputs (same seeded init) and, for
    the gradients, vs the oracle's autograd on the same parameters."""
    from prifit_amd.models import pointnet2_cls_msg, pointnet2_cls_ssg, pointnet2_sem_seg
    g = golden("model_variants")
    xyz = _t(synth.cloud("surface", 2, 1024, int(g["seed"]))).transpose(1, 2).contiguous()
    starts = [_t(g["s%d" % i]) for i in range(4)]
    for name, mod, msg in (("cls_msg", pointnet2_cls_msg, True), ("cls_ssg", pointnet2_cls_ssg, False)):
        torch.manual_seed(5)
        net = mod.get_model(40, normal_channel=False)
        o = orc.OracleCls(40, normal_channel=False, msg=msg)
        o.load_state_dict(net.state_dict())
        for m in (net, o):
            m.train(); m.drop1.eval(); m.drop2.eval()
        net.cuda()
        lp, l3 = net(xyz.cuda(), fps_start=(starts[0].cuda(), starts[1].cuda()))
        assert lp.shape == (2, 40) and l3.shape == (2, 1024, 1)
        torch.testing.assert_close(l3.detach().sum(dim=1).cpu(), _t(g[name + "_l3_sum"]), rtol=1e-3, atol=5e-2)
        # the head's BatchNorm runs on a batch of 2, which amplifies rounding: loose on the log-probs
        torch.testing.assert_close(lp.detach().cpu(), _t(g[name + "_logp"]), rtol=2e-2, atol=2e-2)
        olp, _ = o(xyz, fps_start=(starts[0], starts[1]))
        tgt = torch.tensor([3, 17])
        mod.get_loss()(lp, tgt.cuda()).backward()
        torch.nn.functional.nll_loss(olp, tgt).backward()
        for (n_, p), (_, q) in zip(net.named_parameters(), o.named_parameters()):
            # only the classifier layer: the head's BatchNorm1d sees a batch of TWO rows, where (x - mean) / std is
            # +-1 whatever x is, so every gradient upstream of it is rounding noise amplified (norms ~1e3)
            if n_.startswith("fc3"):
                ref = q.grad
                assert (p.grad.cpu() - ref).norm() <= 5e-2 * ref.norm() + 1e-6, n_
    feats = _t(synth.features(2, 1024, 3, 13)).transpose(1, 2)
    x6 = torch.cat([xyz, feats], 1).contiguous()
    torch.manual_seed(6)
    net = pointnet2_sem_seg.get_model(13, with_rgb=True)
    o = orc.OracleSemSeg(13, with_rgb=True)
    o.load_state_dict(net.state_dict())
    for m in (net, o):
        m.train(); m.drop1.eval()
    net.cuda()
    lp, l4 = net(x6.cuda(), fps_start=tuple(s.cuda() for s in starts))
    assert lp.shape == (2, 1024, 13) and l4.shape == (2, 512, 16)
    torch.testing.assert_close(l4.detach().cpu(), _t(g["sem_l4"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(lp.detach().sum(dim=1).cpu(), _t(g["sem_logp_sum"]), rtol=1e-3, atol=5e-2)
    olp, _ = o(x6, fps_start=tuple(starts))
    tgt = torch.from_numpy(synth.labels(2, 1024, 13, 14)) if hasattr(synth, "labels") else torch.zeros(2, 1024, dtype=torch.long)
    w = torch.linspace(0.5, 1.5, 13)
    pointnet2_sem_seg.get_loss()(lp.reshape(-1, 13), tgt.reshape(-1).cuda(), None, w.cuda()).backward()
    torch.nn.functional.nll_loss(olp.reshape(-1, 13), tgt.reshape(-1), weight=w).backward()
    for (n_, p), (_, q) in zip(net.named_parameters(), o.named_parameters()):
        if n_ in ("conv2.weight", "fp1.mlp_convs.0.weight", "sa4.mlp_convs.2.weight", "sa1.mlp_convs.0.weight"):
            ref = q.grad
            assert (p.grad.cpu() - ref).norm() <= 5e-2 * ref.norm() + 1e-6, n_


@pytest.mark.gpu
def test_sample_ahead_matches_inline_sampling():
    """ops.sample_ahead (the next batch's farthest-point-sampling chain on a side stream) hands out the indices and
    coordinates the in-line launches produce, and a network step fed with them is the same step bit for bit."""
    from prifit_amd import ops, synth
    from prifit_amd.models import pointnet2_part_seg_msg as M
    B, N = 4, 2048
    xyz_cl = _t(synth.cloud("surface", B, N, 21)).cuda()
    s1, s2 = _t(synth.fps_start(B, N, 5)).cuda(), _t(synth.fps_start(B, 512, 6)).cuda()
    a1, a2 = ops.sample_ahead(xyz_cl, (512, 128), (s1, s2))
    i1, x1 = ops.farthest_point_sample(xyz_cl, 512, s1, return_xyz=True)
    i2, x2 = ops.farthest_point_sample(x1, 128, s2, return_xyz=True)
    j1, y1 = ops.farthest_point_sample(xyz_cl, 512, a1, return_xyz=True)   # waits for the side stream's event
    j2, y2 = ops.farthest_point_sample(y1, 128, a2, return_xyz=True)
    assert torch.equal(i1, j1) and torch.equal(i2, j2) and torch.equal(x1, y1) and torch.equal(x2, y2)
    with pytest.raises(ValueError):
        ops.farthest_point_sample(xyz_cl, 256, a1)

    torch.manual_seed(0)
    net = M.get_model(50)
    synth.xavier_like_trainer(net)
    net.cuda().train()
    net.drop1.eval()
    xyz = xyz_cl.transpose(1, 2).contiguous()
    cls = torch.zeros(B, 1, 16, device="cuda")
    outs = []
    for starts in ((s1, s2), net.sample_ahead(xyz, (s1, s2))):
        for p in net.parameters():
            p.grad = None
        seg = net(xyz, cls, fps_start=starts)[0]
        seg.square().mean().backward()
        outs.append((seg.detach().clone(), net.sa1.conv_blocks[0][0].weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-7)   # (float atomics in the backward)


def test_col_sum_is_deterministic_and_exact_enough(hiplib):
    """prifit_col_sum (bias gradient of a convolution without BatchNorm): per-workgroup partial rows added in a fixed order --
    the same bits from run to run (round 4 added them with float atomics), and the value of an fp64 sum to fp32 rounding."""
    import ctypes
    from prifit_amd._lib import call, cur_stream, dll, ptr
    for P, C in ((49152, 128), (5000, 52), (300, 4)):
        Y = torch.randn(P, C, device="cuda") * 3.0 + 0.5
        outs = []
        for _ in range(3):
            out = torch.empty(C, device="cuda")
            ws = torch.empty(dll().prifit_col_sum_workspace(P, C), device="cuda")
            call("prifit_col_sum", ptr(Y), ctypes.c_longlong(C), P, C, ptr(out), ptr(ws), cur_stream())
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        want = Y.double().sum(0).cpu()
        assert (outs[0].double() - want).abs().max() <= 1e-5 * Y.double().abs().sum(0).max().cpu()


@pytest.mark.parametrize("P,C", [(49152, 50), (1000, 16), (4097, 64), (7, 3)])
def test_cross_entropy_matches_torch(nn_ops, P, C):
    """nn_ops.cross_entropy (the segmentation loss of models/pointnet2_part_seg_msg.py:137-144 on the GPU, one pass each way)
    against F.cross_entropy on the same rows in fp64: value and gradient, on log-probabilities as upstream feeds it and on raw
    scores; the same bits from run to run."""
    g = torch.Generator().manual_seed(P + C)
    raw = torch.randn(P, C, generator=g) * 3.0
    target = torch.randint(0, C, (P,), generator=g)
    for x in (torch.log_softmax(raw, dim=1), raw):
        xr = x.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(xr, target)
        ref.backward()
        xd = x.cuda().requires_grad_(True)
        got = nn_ops.cross_entropy(xd, target.cuda())
        (got * 1.5).backward()
        assert abs(got.item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-7
        torch.testing.assert_close(xd.grad.cpu().double(), 1.5 * xr.grad, rtol=1e-5, atol=1e-9)
        xd2 = x.cuda().requires_grad_(True)
        got2 = nn_ops.cross_entropy(xd2, target.cuda())
        (got2 * 1.5).backward()
        assert torch.equal(got, got2) and torch.equal(xd.grad, xd2.grad)


def test_cross_entropy_ignore_index_and_invalid_labels(nn_ops):
    """torch's default label semantics in nn_ops.cross_entropy: rows labelled -100 (ignore_index) contribute nothing
    and are left out of the mean's denominator -- value and gradient against F.cross_entropy in fp64 --; any other label outside
    [0, C) is an error: torch stops with a device-side assert, here the loss and that row's gradient are NaN."""
    P, C = 5000, 50
    g = torch.Generator().manual_seed(7)
    x = torch.randn(P, C, generator=g) * 2.0
    target = torch.randint(0, C, (P,), generator=g)
    target[torch.randperm(P, generator=g)[:700]] = -100
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, target)
    ref.backward()
    xd = x.cuda().requires_grad_(True)
    got = nn_ops.cross_entropy(xd, target.cuda())
    got.backward()
    assert abs(got.item() - ref.item()) <= 2e-6 * abs(ref.item())
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-10)
    assert float(xd.grad[target.cuda() == -100].abs().max()) == 0.0
    all_ignored = nn_ops.cross_entropy(x.cuda(), torch.full((P,), -100, dtype=torch.long, device="cuda"))
    assert torch.isnan(all_ignored)                                   # 0 / 0, as torch
    for bad in (C, -1, 10 ** 6):
        t2 = target.clone()
        t2[17] = bad
        xd = x.cuda().requires_grad_(True)
        loss = nn_ops.cross_entropy(xd, t2.cuda())
        loss.backward()
        assert torch.isnan(loss) and bool(torch.isnan(xd.grad[17]).all())
        ok = torch.ones(P, dtype=torch.bool); ok[17] = False
        assert bool(torch.isfinite(xd.grad[ok.cuda()]).all())
