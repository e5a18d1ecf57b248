"""Autograd functions of the shared per-position MLP stack, built on the C-ABI kernels.

Activations are channels-last matrices [P, C] (P = B*S*K grouped samples or B*N points).  A
"shared MLP" is the reference's (Conv1x1 -> train-mode BatchNorm -> ReLU) x L chain, optionally
followed by the max over the K samples of each group (models/pointnet_util.py:195-199,
:252-256, :310-313).  Here each layer is ONE MFMA GEMM whose prologue applies the previous
layer's BatchNorm+ReLU while loading ("normalise on load") and whose epilogue emits the partial
sums for its own BatchNorm, so no normalised activation is ever written to HBM.
"""
import ctypes
import os

import torch

from . import profiler
from . import arena as zero_pool
from ._lib import call, cur_stream, dll, ptr, query

NT, NN, TN = 0, 1, 2
EPI_NONE, EPI_CHORD, EPI_MSKERNEL, EPI_MSBWD = 0, 1, 2, 3
_LL = ctypes.c_longlong
_F = ctypes.c_float
_D = ctypes.c_double


def _rows_per_slab():
    return query("prifit_reduce_rows_per_slab")


# BatchNorm tails (round 6, include/prifit_hip.h: prifit_bn_fwd / prifit_bn_bwd): the launch that produces a
# layer's column sums also finalizes them -- fp64 atomics into 32 replicas of a [2][C] accumulator, the workgroup that draws the
# last ticket writes the coefficients -- instead of writing a slab per workgroup for a separate prifit_bn_finalize /
# prifit_bn_bwd_finalize launch.  MEASURED AND NOT THE DEFAULT (profiles/r06_bn_tail.txt, same box, alternating runs): 50
# launches less per c3 step (317 -> 267), host enqueue -0.15 ms, the two finalize families' 0.33 ms of GPU time gone -- and
# +0.8 ms in the producers: c3 15.10 -> 15.59 ms per step.  Atomics of many workgroups on one address are performed one after
# the other at ~0.3 us each (memory-side, past the per-XCD L2s), so every producer pays (workgroups / 32) x 0.3 us plus the
# ticket's and the last workgroup's load round trips: +5-9 us for the persistent kernels (break-even against a ~6 us finalize
# launch), +10-30 us for the products with thousands of tiles and the reduce passes.  Same coefficients bit for bit (tested).
# 1: tails on (fewer launches: a host-bound multi-rank job may prefer them).
_BN_TAIL = os.environ.get("PRIFIT_BN_TAIL", "0") != "0"


class _BnFwdDesc(ctypes.Structure):
    _fields_ = [("acc", ctypes.c_void_p), ("ticket", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("count", ctypes.c_double), ("eps", ctypes.c_float), ("momentum", ctypes.c_float)]


class _BnBwdDesc(ctypes.Structure):
    _fields_ = [("acc", ctypes.c_void_p), ("ticket", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("mean", ctypes.c_void_p),
                ("invstd", ctypes.c_void_p), ("out", ctypes.c_void_p), ("count", ctypes.c_double), ("training", ctypes.c_int)]


_REPLICAS = None


def _tail_replicas():
    global _REPLICAS
    if _REPLICAS is None:
        _REPLICAS = query("prifit_bn_tail_replicas")
    return _REPLICAS


def bn_fwd_tail(C, count, gamma, beta, rmean, rvar, eps, momentum, dev):
    """-> (descriptor for the producing launch, coef [4, C] = scale, shift, mean, invstd once that launch has run)."""
    R = _tail_replicas()
    st = zero_pool.zeros(4 * C * R + 4, device=dev)        # [R][2][C] doubles + the ticket, zero on entry
    coef = torch.empty(4, C, dtype=torch.float32, device=dev)
    p = st.data_ptr()
    d = _BnFwdDesc(p, p + 16 * C * R, gamma.data_ptr(), beta.data_ptr(), None if rmean is None else rmean.data_ptr(),
                   None if rvar is None else rvar.data_ptr(), coef.data_ptr(), float(count), float(eps), float(momentum))
    d._keep = (st, coef, gamma, beta, rmean, rvar)
    return d, coef


def bn_bwd_tail(C, count, scale, mean, invstd, training, dev):
    """-> (descriptor, coef [5, C] = dgamma, dbeta, a, b, d of dY = a Gm + b Y + d once the producing launch has run)."""
    R = _tail_replicas()
    st = zero_pool.zeros(4 * C * R + 4, device=dev)
    coef = torch.empty(5, C, dtype=torch.float32, device=dev)
    p = st.data_ptr()
    d = _BnBwdDesc(p, p + 16 * C * R, scale.data_ptr(), mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), float(count), int(training))
    d._keep = (st, coef, scale, mean, invstd)
    return d, coef


def _ref(d):
    return None if d is None else ctypes.byref(d)


_STREAM = os.environ.get("PRIFIT_GEMM_STREAM", "1") != "0"   # 0: every product takes the tiled kernel (A/B runs)
_FUSE_RED = os.environ.get("PRIFIT_FUSE_BN_REDUCE", "1") != "0"  # 0: separate bn_relu_bwd_reduce launches (A/B runs)
_FUSE_POOL_FWD = os.environ.get("PRIFIT_FUSE_POOL_FWD", "1") != "0"  # 0: pool_fwd re-reads the last layer's Y (A/B runs)
_FUSE_BN_APPLY = os.environ.get("PRIFIT_FUSE_BN_APPLY", "1") != "0"  # 0: bn_relu_bwd_apply writes a middle layer's dY (A/B runs)
_FUSE_POOL = os.environ.get("PRIFIT_FUSE_POOL_BWD", "1") != "0"  # 0: pool_bwd_apply writes the pooled layer's dY (A/B runs)
# dA and dW of a streaming-shape layer from ONE pass over its rows (csrc/gemm_stream_bwd.hip) instead of the separate
# streaming dA (NN) and dW (TN) kernels, which each read G, Y and the previous layer's pre-activation.  "auto" (default):
# where it measured faster -- the unpooled middle layers (_FUSE_BWD_AUTO below; round 3 had only the 96 -> 64 one);
# "1": every supported shape (slower on the others: the kernel's dW role is latency-bound); "0": never.
_FUSE_BWD = os.environ.get("PRIFIT_FUSE_DA_DW", "auto")
# (round 5 carried an opt-in algebraic form of the max-pooled set-abstraction layers' backward here, PRIFIT_POOL_ALG: parity-green,
# its dense pass 30-55 % faster than the dA / dW pair, slower on the step because of the winners' index work; removed in round 6,
# the measurements are in DESIGN.md Appendix A.  The layer pooled over the WHOLE cloud keeps that form: src/dgcnn.py.)

# (round 4, tools/fam_table.py on one box: one-pass kernel against the separate dA + dW pair)
#   [1.57 M x 96 x 64] 546 / 712 us, [786 K x 64 x 64] 171 / 282, [197 K x 128 x 128] 152 / 176, [49 K x 128 x 128] 54 / 72;
#   pooled layers stay on the pairs: [1.57 M x 128 x 96] 887 / 874, [786 K x 128 x 64] 337 / 304
_FUSE_BWD_AUTO = {(96, 64), (64, 64), (128, 128)}


def _fuse_bwd_on(Cout, Kin, pooled):
    if _FUSE_BWD == "auto":
        return (Cout, Kin) in _FUSE_BWD_AUTO and not pooled
    return _FUSE_BWD not in ("0", "", False)


def _stream_ok(layout, M, N, K, batch=1, splitk=1, epi=EPI_NONE, b_affine=None, a_rowsum=None, accumulate=False,
               aux=None, row_add=None, bias_stride=0):
    """The tall-and-skinny products of the shared MLPs (forward NT, dA NN) take the weights-stationary streaming
    kernel (csrc/gemm_stream.hip); everything else the tiled kernel (csrc/gemm.hip)."""
    return (_STREAM and batch == 1 and splitk == 1 and epi == EPI_NONE and b_affine is None and a_rowsum is None and
            not accumulate and aux is None and row_add is None and
            bool(query("prifit_gemm_stream_supported", layout, M, N, K)))


def slab_sum(part):
    """part [nslab, ...] -> its sum over the slabs in a fixed order (prifit_slab_sum; torch's reduction: 10-13 us + a memset)."""
    part = part.contiguous()
    out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    call("prifit_slab_sum", ptr(part), part.shape[0], _LL(out.numel()), ptr(out), cur_stream())
    return out


def gemm_stats_slabs(M, N, K):
    """Number of column-statistics slabs a forward (NT) product of this shape writes."""
    if _stream_ok(NT, M, N, K):
        return query("prifit_gemm_stream_slabs", M, K)
    t = query("prifit_gemm_stats_tile_m", M, N)
    return (M + t - 1) // t


def gemm(layout, M, N, K, A, lda, B, ldb, C, ldc, batch=1, sA=0, sB=0, sC=0, splitk=1, a_affine=None,
         b_affine=None, bias=None, bias_stride=0, stats=None, epi=EPI_NONE, epi_scalar=None, aux=None, ld_aux=0, s_aux=0,
         row_add=None, a_rowsum=None, accumulate=None, tiled_stats=False, bn=None):
    """tiled_stats=True: the caller reads `stats` as one slab per 128 rows of C (per-sample statistics, src/dgcnn.py);
    otherwise the slab count is gemm_stats_slabs(M, N, K).  bn: a bn_fwd_tail descriptor -- the launch finalizes the column
    statistics itself (stats may then be None)."""
    if accumulate is None:
        accumulate = splitk > 1
    if (not (tiled_stats and stats is not None) and _stream_ok(layout, M, N, K, batch, splitk, epi, b_affine, a_rowsum, accumulate, aux, row_add) and lda % 4 == 0 and
            A.data_ptr() % 16 == 0 and (layout == NN or (ldb % 4 == 0 and B.data_ptr() % 16 == 0))):
        # HBM-bound: the span's work is the algorithmic bytes (A read once, C written once, B once)
        with profiler.span(profiler.tag("gemm_stream_%s" % ("nt", "nn")[layout], M, N, K), 4.0 * (M * K + M * N + N * K)):
            call("prifit_gemm_stream_f32", layout, M, N, K, ptr(A), _LL(lda), ptr(B), _LL(ldb), ptr(C), _LL(ldc),
                 ptr(a_affine[0]) if a_affine else None, ptr(a_affine[1]) if a_affine else None, ptr(bias), ptr(stats),
                 _ref(bn), cur_stream())
        return
    # span name = the kernel instantiation (layout, BN tile) so that it lines up with rocprofv3's per-kernel rows
    with profiler.span(profiler.tag("gemm_%s_bn%d" % (("nt", "nn", "tn")[layout], 32 if N <= 32 else (64 if N <= 64 else (96 if N <= 96 else 128))),
                                    M, N, K, batch, splitk), 2.0 * M * N * K * batch):
        call("prifit_gemm_f32", layout, M, N, K, ptr(A), _LL(lda), _LL(sA), ptr(B), _LL(ldb), _LL(sB), ptr(C),
             _LL(ldc), _LL(sC), batch, splitk,
             ptr(a_affine[0]) if a_affine else None, ptr(a_affine[1]) if a_affine else None,
             ptr(b_affine[0]) if b_affine else None, ptr(b_affine[1]) if b_affine else None,
             ptr(bias), _LL(bias_stride), ptr(stats), epi, ptr(epi_scalar), ptr(aux), _LL(ld_aux), _LL(s_aux), ptr(row_add),
             ptr(a_rowsum), int(accumulate), _ref(bn), cur_stream())


# split-K of the tiled dW products: one round of resident workgroups (512: two 8-wave workgroups per CU) measured best
# ([256 x 196 x 393 K]: 549 / 455 / 466 / 494 us at 256 / 512 / 1024 / 2048 workgroups)
_SPLITK_WGS = 512


def _splitk_for(P, tiles):
    """Workgroups along the reduction for dW = dY^T A: aim at _SPLITK_WGS workgroups, >= 8 k-tiles each."""
    ktiles = (P + 31) // 32
    want = max(1, _SPLITK_WGS // max(1, tiles))
    return int(max(1, min(want, ktiles // 8 if ktiles >= 8 else 1, 4096)))


def _weight_grad(dY, P, Cout, Ain, Kin, a_affine, out=None):
    """dW [Cout, Kin] = dY[P, Cout]^T . A[P, Kin] (A optionally normalised on load).  `out`: a zero-filled
    [Cout, Kin] destination (several layers share one zeroed arena: one fill instead of one per layer)."""
    if _STREAM and query("prifit_gemm_stream_tn_supported", Cout, Kin, P):
        # tall reduction, small output: the LDS-free streaming kernel (csrc/gemm_stream.hip), HBM-bound
        dW = out if out is not None else zero_pool.zeros(Cout, Kin, device=dY.device)
        ws = torch.empty(query("prifit_gemm_stream_tn_workspace", Cout, Kin, P), dtype=torch.float32, device=dY.device)
        with profiler.span(profiler.tag("gemm_stream_tn", Cout, Kin, P), 4.0 * P * (Cout + Kin)):
            call("prifit_gemm_stream_tn_f32", Cout, Kin, _LL(P), ptr(dY), _LL(dY.stride(0)), ptr(Ain), _LL(Ain.stride(0)),
                 ptr(dW), _LL(Kin), ptr(a_affine[0]) if a_affine else None, ptr(a_affine[1]) if a_affine else None,
                 ptr(ws), cur_stream())
        return dW
    tiles = ((Cout + 127) // 128) * ((Kin + 127) // 128)
    sk = _splitk_for(P, tiles)
    if out is not None:
        dW = out
    elif sk == 1:
        dW = torch.empty(Cout, Kin, dtype=torch.float32, device=dY.device)
    else:
        dW = zero_pool.zeros(Cout, Kin, device=dY.device)
    gemm(TN, Cout, Kin, P, dY, dY.stride(0), Ain, Ain.stride(0), dW, Kin, splitk=sk, b_affine=a_affine)
    return dW


def _fused_bwd(P, Cout, Kin, G, Y, scale, shift, ca, cb, cd, arg, Ttab, pool_K, W, Yp, aff_p, stats_p, dW, dev, red):
    """Gp, the (m1, m2) sums of the layer below and dW of one layer in one pass (prifit_gemm_stream_bwd_f32).
    red(Cp, ns_fn) -> (slab or None, BatchNorm-tail descriptor or None, fused_red): SharedMLPFn.backward's red_target."""
    rslab, tail_p, fused = red(Kin, lambda: query("prifit_gemm_stream_bwd_slabs", P, Cout, Kin))
    ws = torch.empty(query("prifit_gemm_stream_bwd_workspace", P, Cout, Kin), dtype=torch.float32, device=dev)
    Gp = torch.empty(P, Kin, dtype=torch.float32, device=dev)
    (sc1, sh1), (mu1, is1) = aff_p, stats_p
    pooled = arg is not None
    # HBM-bound on the narrow layers: every tensor once (G and Y or Y alone, Yp in, Gp out)
    work = 4.0 * P * ((1 if pooled else 2) * Cout + 2 * Kin)
    with profiler.span(profiler.tag("gemm_stream_bwd", P, Cout, Kin, "pool" if pooled else "bn"), work):
        call("prifit_gemm_stream_bwd_f32", _LL(P), Cout, Kin, ptr(None if pooled else G), ptr(Y), ptr(None if pooled else scale),
             ptr(None if pooled else shift), ptr(None if pooled else ca), ptr(cb), ptr(cd), ptr(arg), ptr(Ttab), int(pool_K or 0),
             ptr(W), _LL(Kin), ptr(Yp), _LL(Yp.stride(0)), ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), ptr(Gp), _LL(Kin), ptr(rslab),
             ptr(dW), _LL(Kin), ptr(ws), _ref(tail_p), cur_stream())
    return Gp, fused


class SharedMLPFn(torch.autograd.Function):
    """(conv1x1 + BatchNorm + ReLU) x L [+ max over the K samples of each group].

    apply(x, cfg, *tensors) with, per layer, tensors = (W [Cout, Kin], bias, gamma, beta,
    running_mean, running_var); cfg = dict(pool_K, training, eps, momentum[list]).

    cfg["preact_slab"] (a [nslab, 2, C] column-statistics slab, or True in eval mode) marks x as the
    already-computed pre-activation of layer 0 (GatherLinearFn): layer 0 then has no GEMM and its W/bias slots
    are None; the gradient returned for x is the one w.r.t. that pre-activation.
    cfg["preact_direct"] / cfg["preact_gather"] (training, fused set-abstraction front end): x is plain data and this
    function owns the gradients of the front end's inputs -- direct: the first conv's weight sits in layer 0's W slot;
    gather (first layer by linearity): U sits in layer 0's W slot, the conv's bias in its bias slot and Vc is appended
    behind the 6 L layer tensors; their gradients come out of prifit_gather_linear_bwd_bn with the BatchNorm + ReLU
    backward of layer 0 folded in (no dY of that layer is ever written)."""

    @staticmethod
    def forward(ctx, x, cfg, *tensors):
        L = len(tensors) // 6
        # norows: layer 0's pre-activation rows are not stored (x is None); they are re-formed from (idx, U, Vc) by the
        # three kernels that consume them (cfg["preact_direct"]: norows, U, Vc, idx, P, N, S, K -- pointnet_util)
        nr = cfg.get("preact_direct") if (cfg.get("preact_direct") or {}).get("norows") else None
        if nr is not None:
            P, dev = nr["P"], nr["U"].device
        else:
            P, K0 = x.shape
            dev = x.device
        training = cfg["training"]
        gather = cfg.get("preact_gather") if cfg.get("preact_slab") is not None else None
        tile_m = 128
        Ys, affines, stats_saved, Ws = [], [], [], []
        prev, prev_aff = x, None
        cand = None
        for l in range(L):
            W, b, gamma, beta, rmean, rvar = tensors[6 * l:6 * l + 6]
            preact = l == 0 and cfg.get("preact_slab") is not None
            if preact:
                W, Cout, Kin, Y = None, (nr["U"].shape[-1] if nr is not None else x.shape[1]), 0, x
            else:
                W = W.contiguous()
                Cout, Kin = W.shape
                assert Kin == (nr["U"].shape[-1] if (l == 1 and nr is not None) else prev.shape[1]), (Kin, l)
                Y = torch.empty(P, Cout, dtype=torch.float32, device=dev)
            # the layer's coefficients [4, C] = scale, shift, mean, invstd.  Training: written by the launch that produces the
            # column sums (BatchNorm tail, bn_fwd_tail) or, in the slab form (PRIFIT_BN_TAIL=0), by prifit_bn_finalize
            tail = coef = None
            if training and _BN_TAIL and not (preact and cfg["preact_slab"].dim() == 2):
                tail, coef = bn_fwd_tail(Cout, P, gamma, beta, rmean, rvar, cfg["eps"], cfg["momentum"][l], dev)
            elif training and preact and cfg["preact_slab"].dim() == 2:
                coef = cfg["preact_slab"]        # [4, C]: finalized by the front end's own launch (_sa_group_launch(bn=...))
            elif training:
                coef = torch.empty(4, Cout, dtype=torch.float32, device=dev)
            if coef is not None:
                scale, shift, mean, invstd = coef.unbind(0)

            def finalize(slab, nslab):
                call("prifit_bn_finalize", ptr(slab), nslab, Cout, _D(float(P)), ptr(gamma), ptr(beta),
                     _F(cfg["eps"]), _F(cfg["momentum"][l]), ptr(rmean), ptr(rvar), ptr(scale), ptr(shift),
                     ptr(mean), ptr(invstd), cur_stream())

            if preact:
                if training:
                    slab = cfg["preact_slab"]
                    if slab.dim() == 3:
                        if tail is not None:   # a slab from a front end that did not finalize (functional API paths)
                            tail = None
                        finalize(slab, slab.shape[0])
                else:
                    invstd = torch.rsqrt(rvar + cfg["eps"])
                    mean = rmean.clone()
                    scale = gamma * invstd
                    shift = beta - mean * scale
            elif training and l == 1 and nr is not None:
                # layer 2 on rows that were never stored: the streaming product gathers them from U (prifit_gemm_stream_gather_f32)
                nslab = query("prifit_gemm_stream_slabs", P, Kin)
                slab = None if tail is not None else torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
                with profiler.span(profiler.tag("gemm_stream_nt", P, Cout, Kin, "gather"), 4.0 * (P * Cout + P + Cout * Kin)):
                    call("prifit_gemm_stream_gather_f32", P, Cout, ptr(nr["idx"]), ptr(nr["U"]), ptr(nr["Vc"]), nr["N"], nr["S"],
                         nr["K"], ptr(W), _LL(Kin), ptr(Y), _LL(Cout), ptr(prev_aff[0]), ptr(prev_aff[1]), ptr(b), ptr(slab),
                         _ref(tail), cur_stream())
                if tail is None:
                    finalize(slab, nslab)
            elif training:
                aligned = prev.stride(0) % 4 == 0 and prev.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0
                slab = nslab = None
                if tail is None:
                    tile_m = query("prifit_gemm_stats_tile_m", P, Cout)
                    nslab = gemm_stats_slabs(P, Cout, Kin) if aligned else (P + tile_m - 1) // tile_m
                    slab = torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
                if (l == L - 1 and cfg["pool_K"] and cfg["pool_K"] % 32 == 0 and _FUSE_POOL_FWD and aligned and
                        prev_aff is not None and _stream_ok(NT, P, Cout, Kin)):
                    # max-pooled last layer on the streaming kernel: (max, argmax, min, argmin) per 32 rows and column come
                    # out of the epilogue; the pool below reads those candidates (1/8 of Y) instead of Y
                    cand = torch.empty(P // 32, 4, Cout, dtype=torch.float32, device=dev)
                    with profiler.span(profiler.tag("gemm_stream_nt", P, Cout, Kin, 1), 4.0 * (P * Kin + P * Cout + Cout * Kin)):
                        call("prifit_gemm_stream_pool_f32", P, Cout, Kin, ptr(prev), _LL(prev.stride(0)), ptr(W), _LL(Kin),
                             ptr(Y), _LL(Cout), ptr(prev_aff[0]), ptr(prev_aff[1]), ptr(b), ptr(slab), ptr(cand), _ref(tail), cur_stream())
                elif (l == L - 1 and cfg["pool_K"] and cfg["pool_K"] % 32 == 0 and _FUSE_POOL_FWD and aligned and
                        prev_aff is not None and not _stream_ok(NT, P, Cout, Kin) and query("prifit_gemm_pool_supported", P, Cout, Kin)):
                    # the same on the tiled (persistent) kernel: SA2's 256-wide last layers
                    cand = torch.empty(P // 32, 4, Cout, dtype=torch.float32, device=dev)
                    with profiler.span(profiler.tag("gemm_nt_bn128", P, Cout, Kin, "pool"), 2.0 * P * Cout * Kin):
                        call("prifit_gemm_pool_f32", P, Cout, Kin, ptr(prev), _LL(prev.stride(0)), ptr(W), _LL(Kin), ptr(Y),
                             _LL(Cout), ptr(prev_aff[0]), ptr(prev_aff[1]), ptr(b), ptr(slab), ptr(cand), _ref(tail), cur_stream())
                else:
                    gemm(NT, P, Cout, Kin, prev, prev.stride(0), W, Kin, Y, Cout, a_affine=prev_aff, bias=b, stats=slab, bn=tail)
                if tail is None:
                    finalize(slab, nslab)
            else:
                gemm(NT, P, Cout, Kin, prev, prev.stride(0), W, Kin, Y, Cout, a_affine=prev_aff, bias=b)
                invstd = torch.rsqrt(rvar + cfg["eps"])
                mean = rmean.clone()
                scale = gamma * invstd
                shift = beta - mean * scale
            Ys.append(Y)
            Ws.append(W)
            affines.append((scale, shift))
            stats_saved.append((mean, invstd))
            prev, prev_aff = Y, (scale, shift)
        CL = Ys[-1].shape[1]
        pool_K = cfg["pool_K"]
        arg = None
        if pool_K:
            G = P // pool_K
            # pool_out: a [G, CL] column window of the level's concatenated output (the scales of a multi-scale level write
            # side by side: no torch.cat); the pool kernels take a leading dimension
            out = cfg.get("pool_out")
            if out is None:
                out = torch.empty(G, CL, dtype=torch.float32, device=dev)
            elif tuple(out.shape) != (G, CL) or out.stride(1) != 1 or out.dtype != torch.float32 or out.device != dev:
                raise ValueError("pool_out must be a [G, C] fp32 column window on the input's device")
            arg = torch.empty(G, CL, dtype=torch.int32, device=dev)
            if cand is not None:
                call("prifit_pool_from_candidates", ptr(cand), ptr(prev_aff[0]), ptr(prev_aff[1]), G, pool_K, CL, 0, _F(0.0),
                     ptr(out), _LL(out.stride(0)), ptr(arg), None, cur_stream())
            else:
                call("prifit_pool_fwd", ptr(Ys[-1]), _LL(CL), ptr(prev_aff[0]), ptr(prev_aff[1]), G, pool_K, CL, 0, _F(0.0),
                     ptr(out), _LL(out.stride(0)), ptr(arg), cur_stream())
        else:
            out = torch.empty(P, CL, dtype=torch.float32, device=dev)
            call("prifit_affine_relu", ptr(Ys[-1]), _LL(CL), ptr(prev_aff[0]), ptr(prev_aff[1]), P, CL, 0, _F(0.0),
                 ptr(out), _LL(CL), cur_stream())
        ctx.cfg = {k: v for k, v in cfg.items() if k not in ("preact_slab", "preact_direct", "preact_gather", "pool_out")}
        ctx.preact_gather = gather
        ctx.preact = cfg.get("preact_slab") is not None
        # direct-mode set-abstraction front end: this function owns the gradient of the first conv's weight (tensors[0],
        # upstream layout [C1, D+3]) and computes it with the BatchNorm backward fused in (prifit_sa_first_layer_dw_bn)
        ctx.preact_direct = cfg.get("preact_direct") if ctx.preact else None
        ctx.L = L
        ctx.P, ctx.dev = P, dev
        assert nr is None or (training and L >= 3 and Ys[0] is None)
        ctx.saved = (x, Ys, Ws, affines, stats_saved, arg)
        ctx.biases = [tensors[6 * l + 1] for l in range(L)]
        return out

    @staticmethod
    def backward(ctx, gout):
        cfg, L = ctx.cfg, ctx.L
        x, Ys, Ws, affines, stats_saved, arg = ctx.saved
        training = cfg["training"]
        P, dev = ctx.P, ctx.dev
        nr = ctx.preact_direct if (ctx.preact_direct or {}).get("norows") else None
        # a pooled stack takes its gradient with any row stride (a column slice of the concatenated multi-scale gradient:
        # every kernel that reads it has a leading dimension) -- no copy; the unpooled paths index rows densely
        if not (cfg["pool_K"] and L > 1 and gout.dim() == 2 and gout.stride(1) == 1 and gout.stride(0) % 4 == 0 and
                gout.data_ptr() % 16 == 0):
            gout = gout.contiguous()
        rps = _rows_per_slab()
        grads = [None] * (6 * L)
        extra_grads = (None,) if ctx.preact_gather is not None else ()
        # one zero-filled arena for every weight gradient (split-K adds into it) and, in training, the (exactly
        # zero) bias gradients of this stack
        wslots, total = {}, 0
        for l in range(L):
            if Ws[l] is None:
                continue
            n = Ws[l].numel() if ctx.needs_input_grad[2 + 6 * l] else 0
            nb = Ws[l].shape[0] if (training and ctx.needs_input_grad[2 + 6 * l + 1]) else 0
            wslots[l] = (total, n, total + n, nb)
            total += n + nb
        arena = zero_pool.zeros(total, device=dev) if total else None
        G_in = gout  # gradient w.r.t. the ReLU output of layer l (or pooled output for the last layer)
        fused_red = None
        for l in range(L - 1, -1, -1):
            Y, W = Ys[l], Ws[l]
            Cout, Kin = W.shape if W is not None else ((nr["U"].shape[-1] if Y is None else Y.shape[1]), 0)
            scale, shift = affines[l]
            mean, invstd = stats_saved[l]
            pooled = (l == L - 1) and cfg["pool_K"]
            # the layer's backward coefficients [5, C] = dgamma, dbeta, a, b, d (dY = a Gm + b Y + d): finalized by the launch that
            # produced the (m1, m2) sums -- the dA product of the layer above, or the reduce pass below -- (BatchNorm tail), or by
            # prifit_bn_bwd_finalize from slabs (PRIFIT_BN_TAIL=0)
            tail = None
            final = fused_red is not None and isinstance(fused_red[0], str)
            if final:
                coef = fused_red[1]
            elif _BN_TAIL and fused_red is None:
                tail, coef = bn_bwd_tail(Cout, P, scale, mean, invstd, training, dev)
            else:
                coef = torch.empty(5, Cout, dtype=torch.float32, device=dev)
            dgamma, dbeta, ca, cb, cd = coef.unbind(0)
            # the pooled last layer: dY = T*[k == arg] + b*Y + d is formed inside the streaming dA / dW kernels from Y
            # itself (no pool_bwd_apply pass writing dY, no reads of it) when both consumers are streaming shapes
            fuse_pool = bool(pooled and _FUSE_POOL and training and l > 0 and W is not None and cfg["pool_K"] % 64 == 0 and Cout != 96 and
                             ctx.needs_input_grad[2 + 6 * l] and _stream_ok(NN, P, Kin, Cout) and
                             query("prifit_gemm_stream_tn_supported", Cout, Kin, P))
            direct0 = l == 0 and ctx.preact_direct is not None
            gather0 = l == 0 and ctx.preact_gather is not None
            # a middle layer on streaming shapes: dY = a*(relu mask)*G + b*Y + d is formed inside its two consumers (the dW and
            # the dA kernel read G and Y instead of dY: no bn_relu_bwd_apply pass writing dY, one read of it less)
            fuse_bn = bool(_FUSE_BN_APPLY and _FUSE_RED and not pooled and not direct0 and training and l > 0 and W is not None and
                           ctx.needs_input_grad[2 + 6 * l] and G_in.stride(0) == Cout and G_in.data_ptr() % 16 == 0 and
                           _stream_ok(NN, P, Kin, Cout) and query("prifit_gemm_stream_tn_supported", Cout, Kin, P))
            dY = None if (fuse_pool or direct0 or gather0 or fuse_bn) else torch.empty(P, Cout, dtype=torch.float32, device=dev)
            slab = nslab = None
            if pooled:
                K = cfg["pool_K"]
                G = P // K
                if tail is None:
                    prs = query("prifit_pool_reduce_groups_per_slab")
                    nslab = (G + prs - 1) // prs
                    slab = torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
                call("prifit_pool_bwd_reduce", ptr(G_in), _LL(G_in.stride(0)), ptr(Y), _LL(Cout), ptr(arg),
                     ptr(scale), ptr(shift), ptr(mean), ptr(invstd), G, K, Cout, 0, _F(0.0), ptr(slab), _ref(tail), cur_stream())
            elif final:
                pass
            elif fused_red is not None:
                slab, nslab = fused_red   # emitted by the dA product of the layer above (prifit_gemm_stream_dgrad_f32)
            else:
                if tail is None:
                    nslab = (P + rps - 1) // rps
                    slab = torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
                call("prifit_bn_relu_bwd_reduce", ptr(G_in), _LL(G_in.stride(0)), ptr(Y), _LL(Cout), ptr(scale),
                     ptr(shift), ptr(mean), ptr(invstd), P, Cout, 0, _F(0.0), ptr(slab), _ref(tail), cur_stream())
            fused_red = None
            if slab is not None:
                call("prifit_bn_bwd_finalize", ptr(slab), nslab, Cout, _D(float(P)), int(training), ptr(scale),
                     ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(ca), ptr(cb), ptr(cd), cur_stream())

            def red_target(Cp, ns_fn):
                """Where a dA product of this layer leaves the (m1, m2) sums of the layer below (Cp channels): -> (slab or None,
                descriptor or None, the `fused_red` value of the next iteration)."""
                if _BN_TAIL:
                    (sc_p, _), (mu_p, is_p) = affines[l - 1], stats_saved[l - 1]
                    d, cf = bn_bwd_tail(Cp, P, sc_p, mu_p, is_p, training, dev)
                    return None, d, ("coef", cf)
                ns = ns_fn()
                rs = torch.empty(ns, 2, Cp, dtype=torch.float32, device=dev)
                return rs, None, (rs, ns)

            if fuse_pool:
                Ttab = torch.empty(G, Cout, dtype=torch.float32, device=dev)
                call("prifit_pool_bwd_table", ptr(G_in), _LL(G_in.stride(0)), ptr(Y), _LL(Cout), ptr(arg), ptr(scale),
                     ptr(shift), ptr(ca), G, K, Cout, _F(0.0), ptr(Ttab), cur_stream())
                wo, wn, bo, bn_ = wslots[l]
                dW = arena[wo:wo + wn].view(Cout, Kin)
                if _fuse_bwd_on(Cout, Kin, True) and _FUSE_RED and query("prifit_gemm_stream_bwd_supported", P, Cout, Kin, K) and Ys[l - 1].stride(0) % 4 == 0:
                    G_prev, fused_red = _fused_bwd(P, Cout, Kin, None, Y, None, None, None, cb, cd, arg, Ttab, K, W, Ys[l - 1],
                                                   affines[l - 1], stats_saved[l - 1], dW, dev, red_target)
                    grads[6 * l] = dW
                    if ctx.needs_input_grad[2 + 6 * l + 1]:
                        grads[6 * l + 1] = arena[bo:bo + bn_]
                    grads[6 * l + 2] = dgamma
                    grads[6 * l + 3] = dbeta
                    G_in = G_prev
                    continue
                ws = torch.empty(query("prifit_gemm_stream_tn_workspace", Cout, Kin, P), dtype=torch.float32, device=dev)
                a_aff = affines[l - 1]
                with profiler.span(profiler.tag("gemm_stream_tn", Cout, Kin, P, 1), 4.0 * P * (Cout + Kin)):
                    call("prifit_gemm_stream_tn_pool_f32", Cout, Kin, _LL(P), ptr(Y), _LL(Cout), ptr(Ys[l - 1]),
                         _LL(Ys[l - 1].stride(0)), ptr(dW), _LL(Kin), ptr(a_aff[0]), ptr(a_aff[1]), ptr(arg), ptr(Ttab),
                         ptr(cb), ptr(cd), K, ptr(ws), cur_stream())
                grads[6 * l] = dW
                if ctx.needs_input_grad[2 + 6 * l + 1]:
                    grads[6 * l + 1] = arena[bo:bo + bn_]
                grads[6 * l + 2] = dgamma
                grads[6 * l + 3] = dbeta
                G_prev = torch.empty(P, Kin, dtype=torch.float32, device=dev)
                bias_dw = torch.mv(W.t(), cd)   # the constant d^T W of every row of dY . W
                rslab = tail_p = fused_next = None
                if _FUSE_RED:
                    rslab, tail_p, fused_next = red_target(Kin, lambda: query("prifit_gemm_stream_slabs", P, Cout))
                (sc1, sh1), (mu1, is1) = affines[l - 1], stats_saved[l - 1]
                with profiler.span(profiler.tag("gemm_stream_nn", P, Kin, Cout, 1), 4.0 * (P * Cout + 2 * P * Kin + Kin * Cout)):
                    call("prifit_gemm_stream_dgrad_pool_f32", P, Kin, Cout, ptr(Y), _LL(Cout), ptr(W), _LL(Kin), ptr(G_prev),
                         _LL(Kin), ptr(bias_dw), ptr(arg), ptr(Ttab), ptr(cb), K, ptr(Ys[l - 1]), _LL(Ys[l - 1].stride(0)),
                         ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), ptr(rslab), _ref(tail_p), cur_stream())
                fused_red = fused_next
                G_in = G_prev
                continue
            if direct0:
                # first layer of a direct-mode set-abstraction scale: dW1 = dY^T [feat | rel] with dY formed on load
                info = ctx.preact_direct
                grads[2] = dgamma
                grads[3] = dbeta
                if ctx.needs_input_grad[2]:
                    Bq, Nq, _ = info["xyz"].shape
                    Sq, Kq, Dq = info["new_xyz"].shape[1], info["K"], info["D"]
                    nblk = int(max(1, min(1024, (P + 1023) // 1024)))
                    part = torch.empty(nblk, Cout, Dq + 3, dtype=torch.float32, device=dev)
                    if nr is not None:
                        with profiler.span("sa_first_layer_dw", 4.0 * P * (Cout + 1)):
                            call("prifit_sa_first_layer_dw_bn_gather", ptr(G_in), ptr(nr["U"]), ptr(nr["Vc"]), ptr(scale), ptr(shift),
                                 ptr(ca), ptr(cb), ptr(cd), ptr(info["idx"]), ptr(info["xyz"]), ptr(info["new_xyz"]),
                                 ptr(info["feat"]), Bq, Nq, Sq, Kq, Cout, Dq, int(info["feat_first"]), nblk, ptr(part), cur_stream())
                    else:
                        with profiler.span("sa_first_layer_dw", 4.0 * P * (2 * Cout + 1)):
                            call("prifit_sa_first_layer_dw_bn", ptr(G_in), ptr(Y), ptr(scale), ptr(shift), ptr(ca), ptr(cb), ptr(cd),
                                 ptr(info["idx"]), ptr(info["xyz"]), ptr(info["new_xyz"]), ptr(info["feat"]), Bq, Nq, Sq, Kq, Cout,
                                 Dq, int(info["feat_first"]), nblk, ptr(part), cur_stream())
                    grads[0] = slab_sum(part)
                if ctx.needs_input_grad[3]:
                    grads[1] = zero_pool.zeros(Cout, device=dev)  # bias in front of a batch-stat BatchNorm
                G_in = None
                break
            if gather0:
                # first layer by linearity: dU / dVc straight from (G, Y) -- BatchNorm + ReLU backward formed on load, the
                # scatter staged in LDS (prifit_gather_linear_bwd_bn)
                info = ctx.preact_gather
                Bq, Nq, Sq, Kq = info["B"], info["N"], info["S"], info["K"]
                grads[2] = dgamma
                grads[3] = dbeta
                if _GATHER_BWD_CSR and query("prifit_gather_linear_bwd_csr_supported", Nq, Cout) and G_in.is_contiguous():
                    # as a gather over the in-edge lists of the points: no atomics, no staging, y1 re-formed from U / Vc (the
                    # layer's rows are not read); the CSR of the ball-query lists is built here, once per level and scale
                    Uq, Vq = info["U"].detach().contiguous(), info["Vc"].detach().contiguous()
                    E = Sq * Kq
                    offs = torch.empty(Bq, Nq + 1, dtype=torch.int32, device=dev)
                    lst, pos, own = (torch.empty(Bq, E, dtype=torch.int32, device=dev) for _ in range(3))
                    wsd = torch.empty(query("prifit_gather_linear_bwd_csr_workspace", Bq, Sq, Kq, Cout), dtype=torch.float64, device=dev)
                    dU = torch.empty(Bq, Nq, Cout, dtype=torch.float32, device=dev)
                    dVc = torch.empty(Bq, Sq, Cout, dtype=torch.float32, device=dev)
                    # bytes: G twice (once per pass), the lists, the tables
                    with profiler.span("gather_linear_bwd", 4.0 * (2.0 * P * Cout + 4.0 * P + 2.0 * (Bq * Nq + Bq * Sq) * Cout)):
                        call("prifit_list_csr", ptr(info["idx"]), Bq, Nq, E, ptr(offs), ptr(lst), ptr(pos), ptr(own), cur_stream())
                        call("prifit_gather_linear_bwd_csr", ptr(G_in), ptr(Uq), ptr(Vq), ptr(ctx.biases[0]), ptr(scale), ptr(shift),
                             ptr(ca), ptr(cb), ptr(cd), ptr(info["idx"]), ptr(offs), ptr(lst), ptr(own), Bq, Nq, Sq, Kq, Cout, ptr(dU),
                             ptr(dVc), ptr(wsd), cur_stream())
                else:
                    dU = zero_pool.zeros(Bq, Nq, Cout, device=dev)
                    dVc = zero_pool.zeros(Bq, Sq, Cout, device=dev)
                    with profiler.span("gather_linear_bwd", 4.0 * (2.0 * P * Cout + P + (Bq * Nq + Bq * Sq) * Cout)):
                        call("prifit_gather_linear_bwd_bn", ptr(G_in), ptr(Y), ptr(scale), ptr(shift), ptr(ca), ptr(cb), ptr(cd),
                             ptr(info["idx"]), Bq, Nq, Sq, Kq, Cout, ptr(dU), ptr(dVc), cur_stream())
                grads[0] = dU
                if ctx.needs_input_grad[3]:
                    grads[1] = zero_pool.zeros(Cout, device=dev)   # bias in front of a batch-stat BatchNorm
                extra_grads = (dVc,)
                G_in = None
                break
            if l == 1 and nr is not None:
                # layer 2 over rows that were never stored: dA + dW + the BatchNorm-backward sums of layer 1 in the one-pass
                # kernel, which re-forms layer 1's pre-activations from U (prifit_gemm_stream_bwd_gather_f32)
                assert fuse_bn, "norows: the streaming backward of layer 2 is required (pointnet_util._norows_scales)"
                wo, wn, bo, bn_ = wslots[l]
                dW = arena[wo:wo + wn].view(Cout, Kin)
                rslab, tail_p, fused_next = red_target(Kin, lambda: query("prifit_gemm_stream_bwd_slabs", P, Cout, Kin))
                ws = torch.empty(query("prifit_gemm_stream_bwd_workspace", P, Cout, Kin), dtype=torch.float32, device=dev)
                G_prev = torch.empty(P, Kin, dtype=torch.float32, device=dev)
                (sc1, sh1), (mu1, is1) = affines[0], stats_saved[0]
                with profiler.span(profiler.tag("gemm_stream_bwd", P, Cout, Kin, "gather"), 4.0 * P * (2 * Cout + Kin)):
                    call("prifit_gemm_stream_bwd_gather_f32", _LL(P), Cout, ptr(G_in), ptr(Y), ptr(scale), ptr(shift), ptr(ca),
                         ptr(cb), ptr(cd), ptr(W), _LL(Kin), ptr(nr["idx"]), ptr(nr["U"]), ptr(nr["Vc"]), nr["N"], nr["S"], nr["K"],
                         ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), ptr(G_prev), _LL(Kin), ptr(rslab), ptr(dW), _LL(Kin), ptr(ws),
                         _ref(tail_p), cur_stream())
                grads[6 * l] = dW
                if ctx.needs_input_grad[2 + 6 * l + 1]:
                    grads[6 * l + 1] = arena[bo:bo + bn_]   # bias in front of a batch-stat BatchNorm: zero gradient
                grads[6 * l + 2] = dgamma
                grads[6 * l + 3] = dbeta
                fused_red = fused_next
                G_in = G_prev
                continue
            if fuse_bn:
                wo, wn, bo, bn_ = wslots[l]
                dW = arena[wo:wo + wn].view(Cout, Kin)
                if _fuse_bwd_on(Cout, Kin, False) and query("prifit_gemm_stream_bwd_supported", P, Cout, Kin, 0) and Ys[l - 1].stride(0) % 4 == 0:
                    G_prev, fused_red = _fused_bwd(P, Cout, Kin, G_in, Y, scale, shift, ca, cb, cd, None, None, 0, W, Ys[l - 1],
                                                   affines[l - 1], stats_saved[l - 1], dW, dev, red_target)
                    grads[6 * l] = dW
                    if ctx.needs_input_grad[2 + 6 * l + 1]:
                        grads[6 * l + 1] = arena[bo:bo + bn_]   # bias in front of a batch-stat BatchNorm: zero gradient
                    grads[6 * l + 2] = dgamma
                    grads[6 * l + 3] = dbeta
                    G_in = G_prev
                    continue
                ws = torch.empty(query("prifit_gemm_stream_tn_workspace", Cout, Kin, P), dtype=torch.float32, device=dev)
                a_aff = affines[l - 1]
                with profiler.span(profiler.tag("gemm_stream_tn", Cout, Kin, P, "bn"), 4.0 * P * (2 * Cout + Kin)):
                    call("prifit_gemm_stream_tn_bn_f32", Cout, Kin, _LL(P), ptr(G_in), ptr(Y), _LL(Cout), ptr(Ys[l - 1]),
                         _LL(Ys[l - 1].stride(0)), ptr(dW), _LL(Kin), ptr(a_aff[0]), ptr(a_aff[1]), ptr(scale), ptr(shift),
                         ptr(ca), ptr(cb), ptr(cd), ptr(ws), cur_stream())
                grads[6 * l] = dW
                if ctx.needs_input_grad[2 + 6 * l + 1]:
                    grads[6 * l + 1] = arena[bo:bo + bn_]   # bias in front of a batch-stat BatchNorm: zero gradient
                grads[6 * l + 2] = dgamma
                grads[6 * l + 3] = dbeta
                G_prev = torch.empty(P, Kin, dtype=torch.float32, device=dev)
                rslab, tail_p, fused_next = red_target(Kin, lambda: query("prifit_gemm_stream_slabs", P, Cout))
                (sc1, sh1), (mu1, is1) = affines[l - 1], stats_saved[l - 1]
                with profiler.span(profiler.tag("gemm_stream_nn", P, Kin, Cout, "bn"), 4.0 * (2 * P * Cout + 2 * P * Kin + Kin * Cout)):
                    call("prifit_gemm_stream_dgrad_bn_f32", P, Kin, Cout, ptr(G_in), ptr(Y), _LL(Cout), ptr(W), _LL(Kin),
                         ptr(G_prev), _LL(Kin), ptr(scale), ptr(shift), ptr(ca), ptr(cb), ptr(cd), ptr(Ys[l - 1]),
                         _LL(Ys[l - 1].stride(0)), ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1), ptr(rslab), _ref(tail_p), cur_stream())
                fused_red = fused_next
                G_in = G_prev
                continue
            if pooled:
                call("prifit_pool_bwd_apply", ptr(G_in), _LL(G_in.stride(0)), ptr(Y), _LL(Cout), ptr(arg),
                     ptr(scale), ptr(shift), ptr(ca), ptr(cb), ptr(cd), G, K, Cout, 0, _F(0.0), ptr(dY), _LL(Cout),
                     cur_stream())
            else:
                call("prifit_bn_relu_bwd_apply", ptr(G_in), _LL(G_in.stride(0)), ptr(Y), _LL(Cout), ptr(scale),
                     ptr(shift), ptr(ca), ptr(cb), ptr(cd), P, Cout, 0, _F(0.0), ptr(dY), _LL(Cout), cur_stream())
            grads[6 * l + 2] = dgamma
            grads[6 * l + 3] = dbeta
            if l == 0 and ctx.preact:
                G_in = dY
                break
            A_in = x if l == 0 else Ys[l - 1]
            a_aff = None if l == 0 else affines[l - 1]
            wo, wn, bo, bn_ = wslots[l]
            if ctx.needs_input_grad[2 + 6 * l]:
                grads[6 * l] = _weight_grad(dY, P, Cout, A_in, Kin, a_aff, out=arena[wo:wo + wn].view(Cout, Kin))
            if ctx.needs_input_grad[2 + 6 * l + 1]:
                # bias in front of a batch-stat BatchNorm has zero gradient; with running stats it is sum(dY)
                grads[6 * l + 1] = arena[bo:bo + bn_] if training else dY.sum(dim=0)
            grads[6 * l + 2] = dgamma
            grads[6 * l + 3] = dbeta
            if l > 0 or ctx.needs_input_grad[0]:
                G_prev = torch.empty(P, Kin, dtype=torch.float32, device=dev)
                if l > 0 and _FUSE_RED and _stream_ok(NN, P, Kin, Cout):
                    # streaming dA product with the BatchNorm-backward column sums of layer l-1 in its epilogue
                    rslab, tail_p, fused_next = red_target(Kin, lambda: query("prifit_gemm_stream_slabs", P, Cout))
                    (sc1, sh1), (mu1, is1) = affines[l - 1], stats_saved[l - 1]
                    with profiler.span(profiler.tag("gemm_stream_nn", P, Kin, Cout, 0), 4.0 * (P * Cout + 2 * P * Kin + Kin * Cout)):
                        call("prifit_gemm_stream_dgrad_f32", P, Kin, Cout, ptr(dY), _LL(Cout), ptr(W), _LL(Kin), ptr(G_prev),
                             _LL(Kin), ptr(Ys[l - 1]), _LL(Ys[l - 1].stride(0)), ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1),
                             ptr(rslab), _ref(tail_p), cur_stream())
                    fused_red = fused_next
                elif l > 0 and _FUSE_RED:
                    # tiled kernel, same epilogue
                    rslab, tail_p, fused_next = red_target(Kin, lambda: (P + query("prifit_gemm_stats_tile_m", P, Kin) - 1) // query("prifit_gemm_stats_tile_m", P, Kin))
                    (sc1, sh1), (mu1, is1) = affines[l - 1], stats_saved[l - 1]
                    with profiler.span(profiler.tag("gemm_nn_bn%d" % (32 if Kin <= 32 else (64 if Kin <= 64 else (96 if Kin <= 96 else 128))),
                                                    P, Kin, Cout, "red"), 2.0 * P * Kin * Cout):
                        call("prifit_gemm_dgrad_bnred_f32", P, Kin, Cout, ptr(dY), _LL(Cout), ptr(W), _LL(Kin), ptr(G_prev),
                             _LL(Kin), ptr(Ys[l - 1]), _LL(Ys[l - 1].stride(0)), ptr(sc1), ptr(sh1), ptr(mu1), ptr(is1),
                             ptr(rslab), _ref(tail_p), cur_stream())
                    fused_red = fused_next
                else:
                    gemm(NN, P, Kin, Cout, dY, Cout, W, Kin, G_prev, Kin)
                G_in = G_prev
            else:
                G_in = None
            del dY
        return (G_in, None) + tuple(grads) + extra_grads


class GatherLinearFn(torch.autograd.Function):
    """First layer of a set-abstraction MLP by linearity (upstream models/pointnet_util.py:243-252):
    conv1([feat_j | xyz_j - c_g]) = U_j - Vc_g + bias with U = [feat | xyz] W1^T per point and
    Vc = c W1x^T per centre.  apply(U [B,N,C], Vc [B,S,C], bias, idx [B,S,K], training)
    -> (Y1 [B*S*K, C], column-statistics slab for the BatchNorm that follows)."""

    @staticmethod
    def forward(ctx, U, Vc, bias, idx, training):
        U, Vc, idx = U.contiguous(), Vc.contiguous(), idx.contiguous()
        B, N, C = U.shape
        _, S, K = idx.shape
        assert Vc.shape == (B, S, C) and idx.dtype == torch.int32, (Vc.shape, idx.dtype)
        P = B * S * K
        Y = torch.empty(P, C, dtype=torch.float32, device=U.device)
        rps = _rows_per_slab()
        slab = torch.empty((P + rps - 1) // rps, 2, C, dtype=torch.float32, device=U.device)
        # algorithmic bytes: read U and Vc once, read idx, write the C-wide grouped pre-activations
        with profiler.span("gather_linear", 4.0 * B * (N * C + S * C + S * K + S * K * C)):
            call("prifit_gather_linear_fwd", ptr(U), ptr(Vc), ptr(bias), ptr(idx), B, N, S, K, C, ptr(Y), ptr(slab),
                 cur_stream())
        ctx.save_for_backward(idx)
        ctx.meta = (B, N, S, K, C, training)
        ctx.mark_non_differentiable(slab)
        return Y, slab

    @staticmethod
    def backward(ctx, gY, _gslab):
        (idx,) = ctx.saved_tensors
        B, N, S, K, C, training = ctx.meta
        gY = gY.contiguous()
        dU = zero_pool.zeros(B, N, C, device=gY.device)
        dVc = torch.empty(B, S, C, dtype=torch.float32, device=gY.device)
        call("prifit_gather_linear_bwd", ptr(gY), ptr(idx), B, N, S, K, C, ptr(dU), ptr(dVc), cur_stream())
        db = None
        if ctx.needs_input_grad[2]:
            # a bias in front of a batch-statistics BatchNorm has zero gradient
            db = zero_pool.zeros(C, device=gY.device) if training else gY.sum(dim=0)
        return dU, dVc, db, None, None


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def sa_group_supported(N, nsamples, widths):
    """Limits of prifit_sa_group_linear_fwd (include/prifit_hip.h)."""
    return (N <= 2048 and 1 <= len(nsamples) <= 4 and sum(nsamples) <= 320 and
            all(16 <= c <= 128 and (c & (c - 1)) == 0 for c in widths))


def _sa_group_launch(mode, xyz, new_xyz, feat, feat_first, radii, nsamples, widths, Ws, Us, Vcs, biases,
                     feat_xyz=False, rows=None, bn=None):
    """One launch: ball query for every radius + the first-layer pre-activations Y_r [B*S*K_r, C_r], their
    BatchNorm column-statistics slabs and the int32 index lists.  rows (gather mode): per radius False = do not store
    Y_r (None is returned for it): index lists and statistics only.
    bn: per radius (gamma, beta, running_mean, running_var, eps, momentum) of the BatchNorm behind the first conv -- the launch
    then FINALIZES the statistics (BatchNorm tail) and the second return value holds, per radius, the coefficients [4, C_r] =
    scale, shift, mean, invstd instead of a slab [nslab, 2, C_r]."""
    import numpy as np

    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    R = len(radii)
    dev = xyz.device
    q = query("prifit_sa_group_queries_per_slab", B, S)
    nslab = B * ((S + q - 1) // q)
    rows = [True] * R if rows is None else list(rows)
    assert mode == 1 or all(rows)
    Ys = [torch.empty(B * S * k, c, dtype=torch.float32, device=dev) if keep else None
          for k, c, keep in zip(nsamples, widths, rows)]
    tails = None
    if bn is not None and _BN_TAIL:
        tails = [bn_fwd_tail(c, B * S * k, g_, b_, rm, rv, eps, mom, dev) for c, k, (g_, b_, rm, rv, eps, mom) in zip(widths, nsamples, bn)]
        slabs = [cf for _, cf in tails]
        bn_arr = (ctypes.POINTER(_BnFwdDesc) * R)(*[ctypes.pointer(d) for d, _ in tails])
    else:
        slabs = [torch.empty(nslab, 2, c, dtype=torch.float32, device=dev) for c in widths]
    idxs = [torch.empty(B, S, k, dtype=torch.int32, device=dev) for k in nsamples]
    r2 = (ctypes.c_float * R)(*[float(np.float32(r ** 2)) for r in radii])  # fp32(radius^2), like ops.ball_query_multi
    ns = (ctypes.c_int * R)(*[int(k) for k in nsamples])
    wd = (ctypes.c_int * R)(*[int(c) for c in widths])
    D = 0 if feat is None else feat.shape[-1]
    # algorithmic bytes (SURVEY.md 8d with the grouped-out term = the C1-wide first-layer rows this launch writes):
    # clouds and centres once, index lists once, rows once, plus the per-point / per-centre projections (gather mode)
    # or the feature table (direct mode)
    work = B * (12.0 * (N + S) + sum(4.0 * S * k * (1 + (c if keep else 0)) for k, c, keep in zip(nsamples, widths, rows)) +
                (sum(4.0 * (N + S) * c for c in widths) if mode == 1 else 4.0 * N * D))
    with profiler.span("sa_group_linear", work):
        call("prifit_sa_group_linear_fwd", ptr(xyz), ptr(new_xyz), B, N, S, R, r2, ns, wd, mode, ptr(feat), D,
             int(feat_first), int(feat_xyz), _ptr_array(Ws) if Ws else None, _ptr_array(Us) if Us else None,
             _ptr_array(Vcs) if Vcs else None, _ptr_array(biases), _ptr_array(Ys), None if tails else _ptr_array(slabs), _ptr_array(idxs),
             bn_arr if tails else None, cur_stream())
    return Ys, slabs, idxs


class SAGroupDirectFn(torch.autograd.Function):
    """Ball query + grouping + first conv of every per-radius MLP in one launch, narrow inputs (upstream
    models/pointnet_util.py:87-107, :127-133 / :243-249, :195-197 / :250-252): y = W_r [feat_j | xyz_j - c] + b_r
    computed from the LDS copy of the cloud, no grouped tensor.  apply(xyz [B,N,3], new_xyz [B,S,3], feat [B,N,D] or
    None (D in 0/3/6, data: no gradient), meta = (radii, nsamples, feat_first, training), W_0, b_0, W_1, b_1, ...)
    with W_r [C_r, D+3] in upstream column order -> (Y_0, slab_0, Y_1, slab_1, ...)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, meta, *tensors):
        radii, nsamples, feat_first, training = meta
        Ws = [w.contiguous() for w in tensors[0::2]]
        bs = [None if b is None else b.contiguous() for b in tensors[1::2]]
        widths = [w.shape[0] for w in Ws]
        # upstream feeds the coordinates themselves as point features (models/pointnet2_part_seg_msg.py:69-75)
        feat_xyz = feat is not None and feat.shape[-1] == 3 and feat.data_ptr() == xyz.data_ptr()
        Ys, slabs, idxs = _sa_group_launch(0, xyz, new_xyz, feat, feat_first, radii, nsamples, widths, Ws, None, None, bs,
                                           feat_xyz=feat_xyz)
        ctx.save_for_backward(xyz, new_xyz, feat, *idxs)
        ctx.meta = (nsamples, widths, feat_first, training, [b is not None for b in bs])
        out = []
        for y, sl in zip(Ys, slabs):
            out += [y, sl]
        ctx.mark_non_differentiable(*slabs)
        return tuple(out)

    @staticmethod
    def backward(ctx, *gouts):
        xyz, new_xyz, feat = ctx.saved_tensors[:3]
        idxs = ctx.saved_tensors[3:]
        nsamples, widths, feat_first, training, has_bias = ctx.meta
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        D = 0 if feat is None else feat.shape[-1]
        grads = []
        for r, (K, C) in enumerate(zip(nsamples, widths)):
            gY = gouts[2 * r]
            dW = db = None
            if gY is not None:
                gY = gY.contiguous()
                P = B * S * K
                nblk = int(max(1, min(1024, (P + 1023) // 1024)))
                part = torch.empty(nblk, C, D + 3, dtype=torch.float32, device=gY.device)
                with profiler.span("sa_first_layer_dw", 4.0 * P * (C + 1)):
                    call("prifit_sa_first_layer_dw", ptr(gY), ptr(idxs[r]), ptr(xyz), ptr(new_xyz), ptr(feat), B, N, S,
                         K, C, D, int(feat_first), nblk, ptr(part), cur_stream())
                dW = slab_sum(part)
                if has_bias[r]:
                    # a bias in front of a batch-statistics BatchNorm has zero gradient
                    db = zero_pool.zeros(C, device=gY.device) if training else gY.sum(dim=0)
            grads += [dW, db]
        return (None, None, None, None) + tuple(grads)


class SAGroupGatherFn(torch.autograd.Function):
    """Same launch for wide inputs, the first layer by linearity: y = U_r[b, j] - Vc_r[b, s] + b_r with
    U = [feat | xyz] W1^T per point and Vc = c W1x^T per centre (see GatherLinearFn).  apply(xyz, new_xyz,
    meta = (radii, nsamples, training), U_0, Vc_0, b_0, U_1, ...) -> (Y_0, slab_0, Y_1, slab_1, ...)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, meta, *tensors):
        radii, nsamples, training = meta
        Us = [u.contiguous() for u in tensors[0::3]]
        Vcs = [v.contiguous() for v in tensors[1::3]]
        bs = [None if b is None else b.contiguous() for b in tensors[2::3]]
        widths = [u.shape[-1] for u in Us]
        Ys, slabs, idxs = _sa_group_launch(1, xyz, new_xyz, None, True, radii, nsamples, widths, None, Us, Vcs, bs)
        ctx.save_for_backward(*idxs)
        ctx.meta = (xyz.shape[0], xyz.shape[1], new_xyz.shape[1], nsamples, widths, training, [b is not None for b in bs])
        out = []
        for y, sl in zip(Ys, slabs):
            out += [y, sl]
        ctx.mark_non_differentiable(*slabs)
        return tuple(out)

    @staticmethod
    def backward(ctx, *gouts):
        idxs = ctx.saved_tensors
        B, N, S, nsamples, widths, training, has_bias = ctx.meta
        grads = []
        for r, (K, C) in enumerate(zip(nsamples, widths)):
            gY = gouts[2 * r]
            dU = dVc = db = None
            if gY is not None:
                gY = gY.contiguous()
                dU = zero_pool.zeros(B, N, C, device=gY.device)
                dVc = torch.empty(B, S, C, dtype=torch.float32, device=gY.device)
                call("prifit_gather_linear_bwd", ptr(gY), ptr(idxs[r]), B, N, S, K, C, ptr(dU), ptr(dVc), cur_stream())
                if has_bias[r]:
                    db = zero_pool.zeros(C, device=gY.device) if training else gY.sum(dim=0)
            grads += [dU, dVc, db]
        return (None, None, None) + tuple(grads)


def _few_rows_splitk(M, N, K):
    """Split of the reduction for a product with at most one row tile (M <= 128: the per-SAMPLE rows of the DGCNN decoder's
    offset, 24 x 512 x 1024) and a deep K: its N / 128 workgroups would each walk all of K serially (88 us for 25 MFLOP).
    Aim at ~256 workgroups, >= 2 k-tiles of 32 each."""
    if M > 128 or K < 256:
        return 1
    tiles = max(1, (N + 127) // 128)
    return int(max(1, min(256 // tiles, K // 64)))


class LinearFn(torch.autograd.Function):
    """Y = X W^T + b on [P, C] rows (a conv1x1 without BatchNorm: conv2 / extra_conv_emb,
    models/pointnet2_part_seg_msg.py:109,128)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        W = W.contiguous()
        P, Kin = x.shape
        Cout = W.shape[0]
        sk = _few_rows_splitk(P, Cout, Kin)
        if sk > 1:   # a handful of rows against a deep reduction: a few workgroups would each walk the whole K serially
            Y = zero_pool.zeros(P, Cout, device=x.device)
            gemm(NT, P, Cout, Kin, x, Kin, W, Kin, Y, Cout, bias=b, splitk=sk)
        else:
            Y = torch.empty(P, Cout, dtype=torch.float32, device=x.device)
            gemm(NT, P, Cout, Kin, x, Kin, W, Kin, Y, Cout, bias=b)
        ctx.save_for_backward(x, W)
        return Y

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gy = gy.contiguous()
        P, Kin = x.shape
        Cout = W.shape[0]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            sk = _few_rows_splitk(P, Kin, Cout)
            if sk > 1:
                dx = zero_pool.zeros(P, Kin, device=x.device)
                gemm(NN, P, Kin, Cout, gy, Cout, W, Kin, dx, Kin, splitk=sk)
            else:
                dx = torch.empty(P, Kin, dtype=torch.float32, device=x.device)
                gemm(NN, P, Kin, Cout, gy, Cout, W, Kin, dx, Kin)
        if ctx.needs_input_grad[1]:
            dW = _weight_grad(gy, P, Cout, x, Kin, None)
        if ctx.needs_input_grad[2]:
            if Cout % 4 == 0 and gy.data_ptr() % 16 == 0:
                db = torch.empty(Cout, dtype=torch.float32, device=x.device)
                ws = torch.empty(query("prifit_col_sum_workspace", P, Cout), dtype=torch.float32, device=x.device)
                call("prifit_col_sum", ptr(gy), _LL(Cout), P, Cout, ptr(db), ptr(ws), cur_stream())
            else:
                db = gy.sum(dim=0)
        return dx, dW, db


class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(x, target) (mean over the rows; models/pointnet2_part_seg_msg.py:137-144) for x [P, C <= 64] on the GPU, with
    torch's default label semantics: -100 (ignore_index) rows are skipped and left out of the denominator, any other label outside
    [0, C) turns the loss into NaN (torch: a device-side assert):
    one pass forward (+ a one-workgroup mean), one pass backward (prifit_cross_entropy_fwd / _bwd).  torch's nll_loss reduces
    with a single workgroup: 73 + 47 us per step at 49152 x 50."""

    @staticmethod
    def forward(ctx, x, target):
        if x.stride(1) != 1:
            x = x.contiguous()
        target = target.contiguous()
        P, C = x.shape
        lse = torch.empty(P, dtype=torch.float32, device=x.device)
        ws = torch.empty(query("prifit_cross_entropy_workspace"), dtype=torch.float32, device=x.device)
        loss = torch.empty(2, dtype=torch.float32, device=x.device)     # (mean over the kept rows, number of kept rows)
        call("prifit_cross_entropy_fwd", ptr(x), _LL(x.stride(0)), ptr(target), _LL(P), C, ptr(lse), ptr(ws), ptr(loss), cur_stream())
        ctx.save_for_backward(x, target, lse, loss)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, target, lse, loss = ctx.saved_tensors
        P, C = x.shape
        dx = torch.empty(P, C, dtype=torch.float32, device=x.device)
        g = g.reshape(1).to(torch.float32).contiguous()
        call("prifit_cross_entropy_bwd", ptr(x), _LL(x.stride(0)), ptr(target), ptr(lse), ptr(g), ptr(loss[1:]), _LL(P), C, ptr(dx),
             _LL(C), cur_stream())
        return dx, None


# SA first layer by linearity, backward as a gather over the points' in-edge lists (0: the walk-and-stage kernel with atomics; A/B, tested)
_GATHER_BWD_CSR = os.environ.get("PRIFIT_GATHER_BWD_CSR", "1") != "0"
# the segmentation loss on the library's kernels (0: torch's F.cross_entropy; A/B arm)
_CE_KERNEL = os.environ.get("PRIFIT_CE_KERNEL", "1") != "0"


def cross_entropy(x, target):
    """F.cross_entropy with the default arguments; rows of up to 64 classes on the GPU go through CrossEntropyFn."""
    if _CE_KERNEL and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] <= 64 and target.dim() == 1 and target.dtype == torch.int64:
        return CrossEntropyFn.apply(x, target)
    return torch.nn.functional.cross_entropy(x, target)


class GroupGatherFn(torch.autograd.Function):
    """Grouped rows [B*S*K, ld] = [feat[idx], xyz[idx] - centre, 0-pad] (order 0) or
    [xyz[idx] - centre, feat[idx], 0-pad] (order 1); gradient flows to `feat` only (xyz is data)."""

    @staticmethod
    def forward(ctx, feat, xyz, new_xyz, idx, order, ld_out):
        from . import ops

        out = ops.group_gather(feat, xyz, new_xyz, idx, order=order, ld_out=ld_out)
        ctx.save_for_backward(idx)
        ctx.meta = (feat.shape, order) if feat is not None else None
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import ops

        if ctx.meta is None or not ctx.needs_input_grad[0]:
            return (None,) * 6
        (idx,) = ctx.saved_tensors
        (B, N, C), order = ctx.meta
        gout = gout.contiguous()
        dfeat = ops.group_scatter_add(gout, 0 if order == 0 else 3, idx, B, N, C)
        return dfeat, None, None, None, None, None


class ConcatWindowsFn(torch.autograd.Function):
    """The concatenation of column windows that were written IN PLACE into one buffer (SharedMLPFn cfg["pool_out"]): returns the
    buffer as a function of the windows -- no copy forward, column slices of the gradient backward.
    apply(wide [R, sum C_i], *windows) -> [R, sum C_i]."""

    @staticmethod
    def forward(ctx, wide, *windows):
        ctx.widths = [w.shape[1] for w in windows]
        return wide.view_as(wide)

    @staticmethod
    def backward(ctx, g):
        outs, c0 = [], 0
        for wd in ctx.widths:
            outs.append(g[:, c0:c0 + wd])
            c0 += wd
        return (None, *outs)


class FpRowsFn(torch.autograd.Function):
    """[interpolated | points1 | 0-pad] rows of a feature-propagation MLP, [B N, kp], in one launch (prifit_fp_rows).
    apply(points2 [B,S,D2], idx [B,N,3] or None (S == 1), weight, points1 [B,N,D1] or None, kp)."""

    @staticmethod
    def forward(ctx, points2, idx, weight, points1, kp):
        points2 = points2.contiguous()
        B, S, D2 = points2.shape
        N = idx.shape[1] if idx is not None else points1.shape[1]
        D1 = 0 if points1 is None else points1.shape[-1]
        p1 = None if points1 is None else points1.contiguous()
        out = torch.empty(B * N, kp, dtype=torch.float32, device=points2.device)
        call("prifit_fp_rows", ptr(points2), ptr(idx), ptr(weight), ptr(p1), B, N, S, D2, D1, kp, ptr(out), cur_stream())
        ctx.save_for_backward(idx, weight)
        ctx.dims = (B, N, S, D2, D1)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        idx, weight = ctx.saved_tensors
        B, N, S, D2, D1 = ctx.dims
        g = g.contiguous()
        if idx is None:      # S == 1: the gradient of the broadcast
            dp2 = g[:, :D2].reshape(B, N, D2).sum(dim=1, keepdim=True)
        else:
            dp2 = ops.three_interpolate_bwd(g, 0, idx, weight, B, S, D2)
        dp1 = g[:, D2:D2 + D1].reshape(B, N, D1) if D1 else None
        return dp2, None, None, dp1, None


class ThreeInterpolateFn(torch.autograd.Function):
    """interpolated[(b,n), :] = sum_j w[b,n,j] * points2[b, idx[b,n,j], :]"""

    @staticmethod
    def forward(ctx, points2, idx, weight):
        from . import ops

        points2 = points2.contiguous()
        ctx.save_for_backward(idx, weight)
        ctx.shape = points2.shape
        return ops.three_interpolate(points2, idx, weight)

    @staticmethod
    def backward(ctx, gout):
        from . import ops

        idx, weight = ctx.saved_tensors
        B, S, C = ctx.shape
        gout = gout.contiguous()
        return ops.three_interpolate_bwd(gout, 0, idx, weight, B, S, C), None, None
