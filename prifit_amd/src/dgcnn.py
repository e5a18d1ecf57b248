"""DGCNN backbone with GroupNorm on the MI355X backend: call surface of the reference's src/dgcnn.py
(`knn :9-27`, `get_graph_feature :74-107`, `DGCNNEncoderGn :149-222`, `DGCNGn :225-267`) --
BASELINE.json configs[4].  Same constructor arguments, parameter names (state_dict compatible) and
outputs: `DGCNGn.forward(points [B,3,N]) -> (embedding [B,N,emb], seg [B,3,N])`.

Every edge convolution is one MFMA GEMM on channels-last edge rows `[x_j - x_i | x_i]`; the GEMM
epilogue emits per-tile column sums from which the per-sample GroupNorm statistics are formed, and
GroupNorm + LeakyReLU + the max over the k neighbours run fused in one kernel (per-sample
coefficient tables).  `get_model(num_part, normal_channel, k)` is the adapter the reference's trainer
expects for `'dgcnn' in args.model` (train_partseg_shapenet.py:226-228) but never shipped.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import arena as zero_pool
from .. import nn_ops
from .. import profiler
from .._lib import call, cur_stream, dll, ptr, query
from ..nn_ops import NN, NT, TN, LinearFn, gemm

_LL = ctypes.c_longlong
_F = ctypes.c_float
_D = ctypes.c_double
# GroupNorm statistics -> coefficient tables in one launch each way (0: the torch fp64 form, ~35 tiny launches per layer; A/B)
_GN_KERNELS = __import__("os").environ.get("PRIFIT_GN_KERNELS", "1") != "0"
# bias / offset gradients of a convolution in front of a GroupNorm from the statistics (0: torch's column sums over dY; A/B arm, tested)
_GN_COLSUMS = __import__("os").environ.get("PRIFIT_GN_COLSUMS", "1") != "0"


def _pad4(c):
    return (c + 3) // 4 * 4


# the first graph (C = 3) straight from the cloud, no [B,N,N] pairwise matrix (prifit_knn3_topk; 0: product + selection, A/B arm)
_KNN3_FUSED = __import__("os").environ.get("PRIFIT_KNN3_FUSED", "1") != "0"


# the pairwise matrix of a feature graph on the symmetric kernel (0: the general product; A/B arm, tested bit-equal)
_KNN_GRAM_SYM = __import__("os").environ.get("PRIFIT_KNN_GRAM_SYM", "1") != "0"


def _knn_cl(x, k):
    """x [B,N,C] channels-last -> idx int32 [B,N,k]."""
    x = x.contiguous()
    B, N, C = x.shape
    if C == 3 and _KNN3_FUSED and query("prifit_knn3_supported", N, k):
        idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
        # a VALU row: per pair one multiply, two fmas and the three operations of the value (8 flop), then the selection
        with profiler.span("knn3_topk", 8.0 * B * N * N):
            call("prifit_knn3_topk", ptr(x), B, N, k, ptr(idx), cur_stream())
        return idx
    G = torch.empty(B, N, N, dtype=torch.float32, device=x.device)
    if _KNN_GRAM_SYM and N % 128 == 0 and C % 32 == 0 and x.data_ptr() % 16 == 0:
        # x x^T is symmetric and so is its arithmetic: the tiles on and above the diagonal, the others as their transposes
        with profiler.span(profiler.tag("gram_sym", N, C, B), 2.0 * B * N * N * C):
            call("prifit_gram_sym_f32", ptr(x), _LL(C), _LL(N * C), ptr(G), _LL(N), _LL(N * N), N, C, B, cur_stream())
    else:
        gemm(NT, N, N, C, x, C, x, C, G, N, batch=B, sA=N * C, sB=N * C, sC=N * N)
    if C == 3:
        xx = (x[..., 0] * x[..., 0] + x[..., 1] * x[..., 1]) + x[..., 2] * x[..., 2]
    else:
        xx = (x * x).sum(dim=-1)
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    with profiler.span("knn_topk", 4.0 * B * N * N + 4.0 * B * N * k):      # HBM: the pairwise matrix once, the lists once
        call("prifit_knn_topk", ptr(G), ptr(xx.contiguous()), B, N, k, ptr(idx), cur_stream())
    return idx


def knn(x, k1, k2):
    """upstream :9-27 -- x [B,C,N] -> idx int64 [B,N,k1] (every (k2//k1)-th of the k2 nearest)."""
    with torch.no_grad():
        idx = _knn_cl(x.transpose(1, 2), k2)
    return idx[:, :, ::k2 // k1].long()


def _knn_normals_cl(x, k):
    """x [B,N,6] (xyz, normals) channels-last -> idx int32 [B,N,k] of upstream :30-71: the k smallest of
    (|p_i|^2 - 2 <p_i, p_j> + |p_j|^2) * (1 + (2 - 2 <n_i, n_j>)).  The two Gram matrices come from the MFMA product (the
    k-ordered fma chain of _knn_cl), the metric is formed with the reference's own operations in its own order, and the
    selection kernel picks the largest of the negated values (G = v / 2 with zero norms makes its (-xx_i + 2 G) - xx_j
    the value v itself, exactly).  First layer of the normals variant only: not on the benchmarked path."""
    B, N, _ = x.shape
    p = F.pad(x[..., 0:3], (0, 1)).contiguous()         # 16-byte rows for the product kernel
    n = F.pad(x[..., 3:6], (0, 1)).contiguous()
    Gp = torch.empty(B, N, N, dtype=torch.float32, device=x.device)
    Gn = torch.empty_like(Gp)
    gemm(NT, N, N, 4, p, 4, p, 4, Gp, N, batch=B, sA=N * 4, sB=N * 4, sC=N * N)
    gemm(NT, N, N, 4, n, 4, n, 4, Gn, N, batch=B, sA=N * 4, sB=N * 4, sC=N * N)
    xx = ((p[..., 0] * p[..., 0] + p[..., 1] * p[..., 1]) + p[..., 2] * p[..., 2]).unsqueeze(1)       # [B,1,N]
    p_pair = xx - 2 * Gp + xx.transpose(2, 1)
    n_pair = 2 - 2 * Gn
    half = (-(p_pair * (1 + n_pair))) * 0.5
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    zero = torch.zeros(B, N, dtype=torch.float32, device=x.device)
    with profiler.span("knn_topk", 4.0 * B * N * N + 4.0 * B * N * k):
        call("prifit_knn_topk", ptr(half.contiguous()), ptr(zero), B, N, k, ptr(idx), cur_stream())
    return idx


def knn_points_normals(x, k1, k2):
    """upstream :30-71 -- x [B,6,N] -> idx int64 [B,N,k1] (every (k2//k1)-th of the k2 nearest under the
    normal-weighted metric)."""
    with torch.no_grad():
        idx = _knn_normals_cl(x.transpose(1, 2).contiguous(), k2)
    return idx[:, :, ::k2 // k1].long()


class EdgeGatherFn(torch.autograd.Function):
    """rows [(b,n,j), ld] = [x_j - x_i, x_i, 0-pad]  (upstream :98-105, channels-last)."""

    @staticmethod
    def forward(ctx, x, idx, ld):
        x = x.contiguous()
        B, N, C = x.shape
        k = idx.shape[2]
        out = torch.empty(B * N * k, ld, dtype=torch.float32, device=x.device)
        call("prifit_edge_gather", ptr(x), ptr(idx), B, N, C, k, ld, ptr(out), cur_stream())
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, C, k)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, C, k = ctx.shape
        g = g.contiguous()
        dx = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        call("prifit_edge_scatter", ptr(g), g.stride(0), ptr(idx), B, N, C, k, ptr(dx), cur_stream())
        return dx, None, None


def get_graph_feature(x, k1=20, k2=20, idx=None):
    """upstream :74-107 -- x [B,C,N] -> (feature [B,2C,N,k1], idx)."""
    B, C, N = x.shape
    xt = x.transpose(1, 2).contiguous()
    if idx is None:
        idx = knn(x, k1, k2)
    rows = EdgeGatherFn.apply(xt, idx.int().contiguous(), _pad4(2 * C))
    return rows[:, :2 * C].reshape(B, N, k1, 2 * C).permute(0, 3, 1, 2), idx


def get_graph_feature_with_normals(x, k1=20, k2=20, idx=None):
    """upstream :110-146 -- x [B,6,N] -> feature [B,12,N,k1] on the graph of knn_points_normals."""
    if idx is None:
        idx = knn_points_normals(x, k1, k2)
    return get_graph_feature(x, k1, k2, idx)[0]


def _gn_tables(slab, Bs, sps, Cout, rows, gamma, beta, cfg, offset=None, chsum=None):
    """Per-sample coefficient tables (scale, shift, mean, invstd), each [Bs, C], from the column-statistics slabs
    [Bs * sps][2][C] of a tensor with `rows` rows per sample (`sps` consecutive slabs belong to one sample).
    chsum (optional, float64 [Bs, C], filled by the kernels): the tensor's column sums per sample -- what the backward needs to
    give the column sums of dY without reading dY (prifit_gn_bwd_finalize)."""
    G, eps = cfg["groups"], cfg["eps"]
    dev = slab.device
    m = float(rows * (Cout // G))
    if offset is not None:
        assert query("prifit_gn_finalize_supported", Cout, G) and offset.shape == (Bs, Cout)
        scale, shift, mean, invstd = (torch.empty(Bs, Cout, dtype=torch.float32, device=dev) for _ in range(4))
        call("prifit_gn_finalize_offset", ptr(slab), Bs, sps, Cout, G, _D(m), ptr(gamma.contiguous()),
             ptr(beta.contiguous()), _D(float(eps)), ptr(offset.contiguous()), _D(float(rows)), ptr(scale), ptr(shift), ptr(mean),
             ptr(invstd), ptr(chsum), cur_stream())
    elif _GN_KERNELS and query("prifit_gn_finalize_supported", Cout, G):
        # per-sample group statistics -> [Bs, C] tables in one launch (the torch form below: ~15 single-workgroup launches)
        scale, shift, mean, invstd = (torch.empty(Bs, Cout, dtype=torch.float32, device=dev) for _ in range(4))
        call("prifit_gn_finalize", ptr(slab), Bs, sps, Cout, G, _D(m), ptr(gamma.contiguous()),
             ptr(beta.contiguous()), _D(float(eps)), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(chsum), cur_stream())
    else:
        assert chsum is None
        sums = slab.view(Bs, sps, 2, Cout).double().sum(dim=1)               # [Bs, 2, C] per-sample column sums
        s1 = sums[:, 0].view(Bs, G, -1).sum(-1) / m
        s2 = sums[:, 1].view(Bs, G, -1).sum(-1) / m
        var = (s2 - s1 * s1).clamp_min(0.0)
        invstd_g = torch.rsqrt(var + eps)
        mean = s1.repeat_interleave(Cout // G, dim=1).float().contiguous()   # [Bs, C] tables
        invstd = invstd_g.repeat_interleave(Cout // G, dim=1).float().contiguous()
        scale = (gamma.unsqueeze(0) * invstd).contiguous()
        shift = (beta.unsqueeze(0) - mean * scale).contiguous()
    return scale, shift, mean, invstd


def _gn_forward(Y, slab, tile, gamma, beta, cfg, offset=None, cand=None, chsum=None, shape=None, ystar=None):
    """GroupNorm statistics from the 128-row (or `tile`-row) column-statistics slabs of Y [P, C] -> per-sample coefficient
    tables, then LeakyReLU [+ max over the pool_K rows of each group].  Returns (out, scale, shift, mean, invstd, arg).
    offset [Bs, C] (optional): the normalised tensor is Y + offset[sample] (prifit_gn_finalize_offset: the tables come out
    relative to Y, so nothing downstream changes)."""
    P, Cout = shape if Y is None else Y.shape          # Y None: the product was not stored (candidates only; `shape` given)
    G, rps, slope, pool_K, eps = cfg["groups"], cfg["rps"], cfg["slope"], cfg["pool_K"], cfg["eps"]
    assert P % rps == 0 and rps % tile == 0 and Cout % G == 0
    Bs = P // rps
    dev = slab.device
    scale, shift, mean, invstd = _gn_tables(slab, Bs, rps // tile, Cout, rps, gamma, beta, cfg, offset, chsum)
    arg = None
    if pool_K:
        Gp = P // pool_K
        out = torch.empty(Gp, Cout, dtype=torch.float32, device=dev)
        arg = torch.empty(Gp, Cout, dtype=torch.int32, device=dev)
        if cand is not None:    # (max, argmax, min, argmin) per 32 rows from the product's epilogue: Y is not read again
            call("prifit_pool_from_candidates", ptr(cand), ptr(scale), ptr(shift), Gp, pool_K, Cout, rps, _F(slope), ptr(out),
                 _LL(Cout), ptr(arg), ptr(ystar), cur_stream())
        else:
            call("prifit_pool_fwd", ptr(Y), _LL(Cout), ptr(scale), ptr(shift), Gp, pool_K, Cout, rps, _F(slope),
                 ptr(out), _LL(Cout), ptr(arg), cur_stream())
    else:
        out = torch.empty(P, Cout, dtype=torch.float32, device=dev)
        call("prifit_affine_relu", ptr(Y), _LL(Cout), ptr(scale), ptr(shift), P, Cout, rps, _F(slope), ptr(out),
             _LL(Cout), cur_stream())
    return out, scale, shift, mean, invstd, arg


def _gn_backward(gout, Y, gamma, scale, shift, mean, invstd, arg, cfg, chsum=None, sums=None):
    """Gradient of _gn_forward w.r.t. Y (written as dY [P, C]), gamma and beta.  chsum (the forward's, _gn_tables) and a dict
    `sums`: sums["dsum"] [Bs, C] = the column sums of dY per sample, sums["db"] [C] = over all rows."""
    gout, ca, cb, cd, dgamma, dbeta = _gn_backward_coefs(gout, Y, gamma, scale, shift, mean, invstd, arg, cfg, chsum, sums)
    P, Cout = Y.shape
    rps, slope, pool_K = cfg["rps"], cfg["slope"], cfg["pool_K"]
    dY = torch.empty(P, Cout, dtype=torch.float32, device=Y.device)
    if pool_K:
        call("prifit_pool_bwd_apply", ptr(gout), _LL(gout.stride(0)), ptr(Y), _LL(Cout), ptr(arg), ptr(scale),
             ptr(shift), ptr(ca), ptr(cb), ptr(cd), P // pool_K, pool_K, Cout, rps, _F(slope), ptr(dY), _LL(Cout),
             cur_stream())
    else:
        call("prifit_bn_relu_bwd_apply", ptr(gout), _LL(gout.stride(0)), ptr(Y), _LL(Cout), ptr(scale), ptr(shift),
             ptr(ca), ptr(cb), ptr(cd), P, Cout, rps, _F(slope), ptr(dY), _LL(Cout), cur_stream())
    return dY, dgamma, dbeta


def _gn_backward_coefs(gout, Y, gamma, scale, shift, mean, invstd, arg, cfg, chsum=None, sums=None):
    """The reduction half of the GroupNorm backward: (gout contiguous, ca, cb, cd [Bs, C], dgamma, dbeta) with
    dY = ca * act'(.) * g + cb * Y + cd.  Y None (the cloud-pooled layer whose product was not stored): sums["ystar"] holds the
    winners' pre-activations, sums["shape"] = (P, Cout)."""
    P, Cout = sums["shape"] if Y is None else Y.shape
    G, rps, slope, pool_K = cfg["groups"], cfg["rps"], cfg["slope"], cfg["pool_K"]
    Bs = P // rps
    dev = gout.device
    gout = gout.contiguous()
    rows = query("prifit_reduce_rows_per_slab")
    if pool_K and pool_K == rps:
        # one pooling group per sample (the global max over a cloud, src/dgcnn.py:197): the per-sample partials are the
        # winners' terms themselves, [Bs, C] numbers -- (sum Gm, sum Gm * yhat) with Gm = act'(.) * gout at the winning row
        Gp = P // pool_K
        ystar = None if sums is None else sums.get("ystar")
        yw = ystar if ystar is not None else torch.gather(Y.view(Gp, pool_K, Cout), 1, arg.long().unsqueeze(1)).squeeze(1)   # [Bs, C]
        gm = torch.where(yw * scale + shift > 0, gout, gout * slope)
        slab = torch.stack([gm, gm * ((yw - mean) * invstd)], dim=1).contiguous()                # [Bs, 2, C]
        nslab = Gp
        if sums is not None:
            sums["gm"] = gm
    elif pool_K:
        Gp = P // pool_K
        rows = query("prifit_pool_reduce_groups_per_slab")
        nslab = (Gp + rows - 1) // rows
        slab = torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
        call("prifit_pool_bwd_reduce", ptr(gout), _LL(gout.stride(0)), ptr(Y), _LL(Cout), ptr(arg), ptr(scale),
             ptr(shift), ptr(mean), ptr(invstd), Gp, pool_K, Cout, rps, _F(slope), ptr(slab), None, cur_stream())
    else:
        nslab = (P + rows - 1) // rows
        slab = torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
        call("prifit_bn_relu_bwd_reduce", ptr(gout), _LL(gout.stride(0)), ptr(Y), _LL(Cout), ptr(scale), ptr(shift),
             ptr(mean), ptr(invstd), P, Cout, rps, _F(slope), ptr(slab), None, cur_stream())
    m = float(cfg.get("count_rows", rps) * (Cout // G))     # count_rows: Y is a per-group table of a tensor with more rows
    ca = scale.contiguous()
    if _GN_KERNELS and query("prifit_gn_finalize_supported", Cout, G):
        cb, cd = (torch.empty(Bs, Cout, dtype=torch.float32, device=dev) for _ in range(2))
        S = torch.empty(Bs, 2, Cout, dtype=torch.float64, device=dev)
        dsum = db = None
        if chsum is not None:
            dsum = torch.empty(Bs, Cout, dtype=torch.float32, device=dev)
            db = torch.empty(Cout, dtype=torch.float32, device=dev)
            sums["dsum"], sums["db"] = dsum, db
        call("prifit_gn_bwd_finalize", ptr(slab), Bs, nslab // Bs, Cout, G, _D(m), ptr(gamma.contiguous()), ptr(mean),
             ptr(invstd), ptr(cb), ptr(cd), ptr(S), ptr(chsum), _D(float(rps)), ptr(dsum), cur_stream())
        dgamma, dbeta = (torch.empty(Cout, dtype=torch.float32, device=dev) for _ in range(2))
        call("prifit_gn_param_grads", ptr(S), Bs, Cout, ptr(dgamma), ptr(dbeta), ptr(dsum), ptr(db), cur_stream())
    else:
        S = slab.view(Bs, nslab // Bs, 2, Cout).double().sum(dim=1)          # [Bs, 2, C]: sum Gm, sum Gm*yhat
        dgamma = S[:, 1].sum(0).float()
        dbeta = S[:, 0].sum(0).float()
        gd = gamma.double().unsqueeze(0)
        m1 = (gd * S[:, 0]).view(Bs, G, -1).sum(-1) / m                      # group means of dyhat, dyhat*yhat
        m2 = (gd * S[:, 1]).view(Bs, G, -1).sum(-1) / m
        rep = Cout // G
        m1c, m2c = m1.repeat_interleave(rep, dim=1), m2.repeat_interleave(rep, dim=1)
        isd, mu = invstd.double(), mean.double()
        cb = (-(isd * isd) * m2c).float().contiguous()
        cd = (-isd * m1c + mu * isd * isd * m2c).float().contiguous()
    return gout, ca, cb, cd, dgamma, dbeta


def pool_product_ok(P, Cout, Kin):
    """prifit_gemm_pool_f32 takes this product (more output tiles than resident workgroups, 16-byte rows)."""
    return Kin % 4 == 0 and bool(query("prifit_gemm_pool_supported", P, Cout, Kin))


def _global_pool_alg_ok(P, rps, Cout, Kin, x):
    return bool(_GLOBAL_POOL_ALG and query("prifit_global_pool_winners_supported", Cout, Kin) and rps % 512 == 0 and
                x.data_ptr() % 16 == 0)


def _global_pool_alg_bwd(x, W, bias, ca, cb, cd, gm, arg, rps, need_dx, need_dW):
    """dx [P, Kin] and dW [Cout, Kin] of conv -> GroupNorm -> ReLU -> max over the whole cloud (upstream :194-197) WITHOUT the
    [P, Cout] tensor dY.  Row by row dY = T [row == winner] + cb * Y + cd with Y = x W^T + bias and per-sample tables
    T = ca * gm, cb, cd [Bs, Cout], so with A_b = diag(cb_b) W and e_b = cd_b + cb_b * bias:
        dx_b = x_b (W^T A_b) + 1 (e_b W)^T + (winners: dx[b, arg[b,c]] += T[b,c] W[c])
        dW   = sum_b A_b (x_b^T x_b) + e_b^T (1^T x_b) + (winners: dW[c] += T[b,c] x[b, arg[b,c]])
    -- per-sample [Kin, Kin] products (6.4 + 3.2 GFLOP each way at B = 24 x 2048, Kin = 256, Cout = 1024) in place of two
    25.8 GFLOP products over dY, and the 403 MB that writing and reading dY moved."""
    P, Kin = x.shape
    Cout = W.shape[0]
    Bs = P // rps
    dev = x.device
    T = (ca * gm).contiguous()
    e = cd if bias is None else torch.addcmul(cd, cb, bias)                      # [Bs, Cout]
    Acat = torch.empty(Cout, Bs, Kin, dtype=torch.float32, device=dev)              # A_b = Acat[:, b, :]
    torch.mul(cb.t().unsqueeze(-1), W.unsqueeze(1), out=Acat)
    dx = dW = None
    if need_dx:
        M = zero_pool.zeros(Bs, Kin, Kin, device=dev)
        gemm(TN, Kin, Kin, Cout, W, Kin, Acat, Bs * Kin, M, Kin, batch=Bs, sA=0, sB=Kin, sC=Kin * Kin, splitk=4)    # W^T A_b
        v = torch.mm(e, W)                                                         # [Bs, Kin]
        dx = torch.empty(P, Kin, dtype=torch.float32, device=dev)
        gemm(NN, rps, Kin, Kin, x, Kin, M, Kin, dx, Kin, batch=Bs, sA=rps * Kin, sB=Kin * Kin, sC=rps * Kin, bias=v, bias_stride=Kin)
    if need_dW:
        G = zero_pool.zeros(Bs, Kin, Kin, device=dev)
        gemm(TN, Kin, Kin, rps, x, Kin, x, Kin, G, Kin, batch=Bs, sA=rps * Kin, sB=rps * Kin, sC=Kin * Kin, splitk=4)  # x_b^T x_b
        s = torch.empty(Bs, Kin, dtype=torch.float32, device=dev)                  # 1^T x_b
        ws = torch.empty(query("prifit_col_sum_workspace", P, Kin), dtype=torch.float32, device=dev)
        call("prifit_col_sum_samples", ptr(x), _LL(Kin), P, Kin, rps, ptr(s), ptr(ws), cur_stream())
        dW = zero_pool.zeros(Cout, Kin, device=dev)
        gemm(NN, Cout, Kin, Bs * Kin, Acat, Bs * Kin, G, Kin, dW, Kin, splitk=32)   # sum_b A_b G_b as ONE product, K = Bs Kin
        dW.addmm_(e.t(), s)
    with profiler.span("global_pool_winners", 4.0 * Bs * Cout * (2.0 * Kin + 2.0)):
        call("prifit_global_pool_winners_f32", Bs, rps, Cout, Kin, ptr(arg), ptr(T), ptr(W), _LL(Kin), ptr(x), _LL(Kin), ptr(dx),
             _LL(Kin), ptr(dW), _LL(Kin), cur_stream())
    return dx, dW


class ConvGNActFn(torch.autograd.Function):
    """conv1x1 (+bias) -> GroupNorm(groups) -> LeakyReLU(slope) [-> max over the K rows of each group].

    x [P, Kin] channels-last rows, `rps` rows per sample (GroupNorm statistics are per sample, upstream
    :157-159,171,231-246).  apply(x, W [Cout,Kin], bias|None, gamma, beta, cfg) with
    cfg = dict(groups, rps, slope, pool_K, eps)."""

    @staticmethod
    def forward(ctx, x, W, bias, gamma, beta, cfg, offset=None):
        """offset [Bs, Cout] (optional, differentiable): a per-sample row added to the convolution's output BEFORE the
        normalisation -- never added to the rows, only folded into the coefficient tables (_gn_forward)."""
        x, W = x.contiguous(), W.contiguous()
        P, Kin = x.shape
        Cout = W.shape[0]
        assert cfg["rps"] % 512 == 0
        dev = x.device
        tile = query("prifit_gemm_stats_tile_m", P, Cout)   # rows per statistics slab (divides rps: 64 or 128)
        nslab = (P + tile - 1) // tile
        slab = torch.empty(nslab, 2, Cout, dtype=torch.float32, device=dev)
        # a bias or an offset in front of the normalisation: their gradients are column sums of dY, which the backward's
        # finalize gives from the statistics alone if the forward keeps the column sums of Y (24 x C numbers)
        chsum = None
        if (bias is not None or offset is not None) and _GN_COLSUMS and _GN_KERNELS and query("prifit_gn_finalize_supported", Cout, cfg["groups"]):
            chsum = torch.empty(P // cfg["rps"], Cout, dtype=torch.float64, device=dev)
        cand = ystar = None
        pooled_tiled = bool(cfg["pool_K"] and cfg["pool_K"] % 32 == 0 and tile == 128 and pool_product_ok(P, Cout, Kin))
        # the layer pooled over the whole cloud, backward in the algebraic form: nothing reads Y but the winners' values, which
        # the pool leaves in `ystar` -- the product is not stored (201 MB at B = 24 x 2048 x 1024)
        nostore = bool(pooled_tiled and _GLOBAL_POOL_NOSTORE and cfg["pool_K"] == cfg["rps"] and chsum is not None and
                       offset is None and _global_pool_alg_ok(P, cfg["rps"], Cout, Kin, x))
        Y = None if nostore else torch.empty(P, Cout, dtype=torch.float32, device=dev)
        if pooled_tiled:
            # the pooled layer on the persistent kernel: its epilogue leaves the per-32-row (max, argmax, min, argmin)
            # candidates, so neither the pool nor an activation pass reads Y again
            cand = torch.empty(P // 32, 4, Cout, dtype=torch.float32, device=dev)
            if nostore:
                ystar = torch.empty(P // cfg["pool_K"], Cout, dtype=torch.float32, device=dev)
            call("prifit_gemm_pool_f32", P, Cout, Kin, ptr(x), _LL(Kin), ptr(W), _LL(Kin), ptr(Y), _LL(Cout), None, None,
                 ptr(bias), ptr(slab), ptr(cand), None, cur_stream())
        else:
            gemm(NT, P, Cout, Kin, x, Kin, W, Kin, Y, Cout, bias=bias, stats=slab, tiled_stats=True)
        out, scale, shift, mean, invstd, arg = _gn_forward(Y, slab, tile, gamma, beta, cfg, offset, cand, chsum, (P, Cout), ystar)
        ctx.ystar = ystar
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.has_offset = offset is not None
        ctx.chsum = chsum
        ctx.bias = bias
        ctx.save_for_backward(x, W, gamma, Y, scale, shift, mean, invstd, *([arg] if arg is not None else []))
        return out

    @staticmethod
    def backward(ctx, gout):
        cfg = ctx.cfg
        x, W, gamma, Y, scale, shift, mean, invstd = ctx.saved_tensors[:8]
        arg = ctx.saved_tensors[8] if len(ctx.saved_tensors) > 8 else None
        P, Kin = x.shape
        Cout = W.shape[0]
        dev = x.device
        sums = {}
        if ctx.ystar is not None:
            sums["ystar"], sums["shape"] = ctx.ystar, (P, Cout)
        if Y is None or (cfg["pool_K"] and cfg["pool_K"] == cfg["rps"] and ctx.chsum is not None and not ctx.has_offset and
                         _global_pool_alg_ok(P, cfg["rps"], Cout, Kin, x)):
            # the layer pooled over the whole cloud: everything from the statistics, the input and B x Cout winners
            gout, ca, cb, cd, dgamma, dbeta = _gn_backward_coefs(gout, Y, gamma, scale, shift, mean, invstd, arg, cfg, ctx.chsum, sums)
            dx, dW = _global_pool_alg_bwd(x, W, ctx.bias, ca, cb, cd, sums["gm"], arg, cfg["rps"], ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1])
            db = sums["db"] if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            return dx, dW, db, dgamma, dbeta, None, None
        dY, dgamma, dbeta = _gn_backward(gout, Y, gamma, scale, shift, mean, invstd, arg, cfg, ctx.chsum, sums)
        dW = nn_ops._weight_grad(dY, P, Cout, x, Kin, None) if ctx.needs_input_grad[1] else None
        doff = None
        if ctx.has_offset and ctx.needs_input_grad[6]:
            # the offset reaches every row of its sample: its gradient is the column sum of dY over the sample
            doff = sums["dsum"] if "dsum" in sums else dY.view(P // cfg["rps"], cfg["rps"], Cout).sum(dim=1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = sums["db"] if "db" in sums else (doff.sum(dim=0) if doff is not None else dY.sum(dim=0))
        else:
            db = None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(P, Kin, dtype=torch.float32, device=dev)
            gemm(NN, P, Kin, Cout, dY, Cout, W, Kin, dx, Kin)
        return dx, dW, db, dgamma, dbeta, None, doff


class GNActFn(torch.autograd.Function):
    """GroupNorm -> LeakyReLU [-> max over pool_K rows] on PRE-ACTIVATION rows Y [P, C] that come with their 128-row
    column-statistics slabs (the by-linearity edge convolution: nn_ops.GatherLinearFn).  apply(Y, slab, gamma, beta, cfg)."""

    @staticmethod
    def forward(ctx, Y, slab, gamma, beta, cfg):
        out, scale, shift, mean, invstd, arg = _gn_forward(Y, slab, query("prifit_reduce_rows_per_slab"), gamma, beta, cfg)
        ctx.cfg = cfg
        ctx.save_for_backward(gamma, Y, scale, shift, mean, invstd, *([arg] if arg is not None else []))
        return out

    @staticmethod
    def backward(ctx, gout):
        gamma, Y, scale, shift, mean, invstd = ctx.saved_tensors[:6]
        arg = ctx.saved_tensors[6] if len(ctx.saved_tensors) > 6 else None
        dY, dgamma, dbeta = _gn_backward(gout, Y, gamma, scale, shift, mean, invstd, arg, ctx.cfg)
        return dY, None, dgamma, dbeta, None


class EdgeConvLinFn(torch.autograd.Function):
    """The by-linearity edge convolution block in one autograd node: y[(i,j)] = U[idx[i,j]] - Vc[i] (gather kernel) ->
    GroupNorm -> LeakyReLU -> max over the k neighbours.  apply(U [B,N,C], Vc [B,N,C], idx [B,N,k] int32, gamma, beta, cfg)
    -> [B*N, C].  Backward: the pooled GroupNorm backward is formed on the fly inside the scatter (prifit_gather_linear_
    bwd_pool): no dY tensor between an apply pass and the scatter."""

    @staticmethod
    def forward(ctx, U, Vc, idx, gamma, beta, cfg):
        U, Vc, idx = U.contiguous(), Vc.contiguous(), idx.contiguous()
        B, N, C = U.shape
        k = idx.shape[2]
        P = B * N * k
        dev = U.device
        Y = torch.empty(P, C, dtype=torch.float32, device=dev)
        rows = query("prifit_reduce_rows_per_slab")
        slab = torch.empty((P + rows - 1) // rows, 2, C, dtype=torch.float32, device=dev)
        # HBM: U and Vc once, the index lists, the C-wide edge pre-activations written once
        with profiler.span("edge_gather_linear", 4.0 * (2.0 * B * N * C + P + P * C)):
            call("prifit_gather_linear_fwd", ptr(U), ptr(Vc), None, ptr(idx), B, N, N, k, C, ptr(Y), ptr(slab), cur_stream())
        out, scale, shift, mean, invstd, arg = _gn_forward(Y, slab, rows, gamma, beta, cfg)
        ctx.cfg, ctx.dims = cfg, (B, N, k, C)
        ctx.save_for_backward(idx, gamma, Y, scale, shift, mean, invstd, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, gamma, Y, scale, shift, mean, invstd, arg = ctx.saved_tensors
        B, N, k, C = ctx.dims
        cfg = ctx.cfg
        gout, ca, cb, cd, dgamma, dbeta = _gn_backward_coefs(gout, Y, gamma, scale, shift, mean, invstd, arg, cfg)
        dU = torch.zeros(B, N, C, dtype=torch.float32, device=Y.device)
        dVc = torch.empty(B, N, C, dtype=torch.float32, device=Y.device)
        # float atomics (one per edge element into dU): priced as bytes -- Y read once, 4 B per atomic, the index lists, dVc
        with profiler.span("edge_scatter_pool", 4.0 * (2.0 * B * N * k * C + B * N * k + 2.0 * B * N * C)):
            call("prifit_gather_linear_bwd_pool", ptr(gout), _LL(gout.stride(0)), ptr(Y), ptr(arg), ptr(scale), ptr(shift), ptr(ca),
                 ptr(cb), ptr(cd), ptr(idx), B, N, N, k, C, cfg["rps"], _F(cfg["slope"]), ptr(dU), ptr(dVc), cur_stream())
        return dU, dVc, None, dgamma, dbeta, None


class EdgeConvTabFn(torch.autograd.Function):
    """The same block with NO per-edge tensor (csrc/edge_conv.hip): one pass over the neighbour lists leaves per-point tables
    (max / min / sum over the neighbours of y, their positions, the GroupNorm column sums); the pooled activation comes from
    the tables once the statistics are final (the activation is monotone in y), and the backward needs the tables, U, the
    centre term and a CSR of the neighbour lists: dU is a gather over a point's in-edges (no atomics).
    apply(U, Vc, idx, csr, gamma, beta, cfg) -> [B*N, C] with U, Vc [B,N,C], csr = edge_csr(idx); or apply(UV, None, ...)
    with UV [B,N,2C] = X [Wa; Wb]^T, the two halves of ONE product (y = U_j - U_i + Vb_i), whose gradient comes back as one
    [B,N,2C] tensor."""

    @staticmethod
    def forward(ctx, U, Vc, idx, csr, gamma, beta, cfg):
        stacked = Vc is None
        U, idx = U.contiguous(), idx.contiguous()
        B, N, ld = U.shape
        C = ld // 2 if stacked else ld
        Vc = U[:, :, C:] if stacked else Vc.contiguous()         # a view: rows of stride ld
        k = idx.shape[2]
        dev = U.device
        pts = query("prifit_edge_points_per_slab")
        ymax, ymin, ysum, ystar, vct, out = (torch.empty(B * N, C, dtype=torch.float32, device=dev) for _ in range(6))
        karg = torch.empty(B * N, C, dtype=torch.int32, device=dev)
        slab = torch.empty(B * (N // pts), 2, C, dtype=torch.float32, device=dev)
        # bytes: the index lists, U / Vc once (the k-fold re-reads of U rows are L2 traffic), five tables written
        with profiler.span("edge_stats", 4.0 * (B * N * k + 7.0 * B * N * C)):
            call("prifit_edge_stats", ptr(U), _LL(ld), ptr(Vc), _LL(ld), int(stacked), ptr(idx), B, N, k, C, ptr(ymax), ptr(ymin),
                 ptr(karg), ptr(ysum), ptr(vct), ptr(slab), cur_stream())
        scale, shift, mean, invstd = _gn_tables(slab, B, N // pts, C, N * k, gamma, beta, cfg)
        call("prifit_edge_pool", ptr(ymax), ptr(ymin), ptr(scale), ptr(shift), B, N, C, _F(cfg["slope"]), ptr(out), _LL(C),
             ptr(ystar), cur_stream())
        ctx.cfg, ctx.dims, ctx.stacked = cfg, (B, N, k, C), stacked
        ctx.save_for_backward(U, vct, idx, *csr, gamma, ystar, ysum, karg, scale, shift, mean, invstd)
        return out

    @staticmethod
    def backward(ctx, gout):
        U, vct, idx, offs, lst, pos, gamma, ystar, ysum, karg, scale, shift, mean, invstd = ctx.saved_tensors
        B, N, k, C = ctx.dims
        # the reduction half of the pooled GroupNorm backward sees only the winners: it is the unpooled reduction over the
        # [B N, C] table of winning pre-activations, with the element count of the full tensor
        cfg = dict(ctx.cfg, rps=N, pool_K=0, count_rows=N * k)
        gout, ca, cb, cd, dgamma, dbeta = _gn_backward_coefs(gout, ystar, gamma, scale, shift, mean, invstd, None, cfg)
        if ctx.stacked:
            dUV = torch.empty(B, N, 2 * C, dtype=torch.float32, device=U.device)
            dU, dVc, ld = dUV, dUV[:, :, C:], 2 * C
        else:
            dU, dVc = (torch.empty(B, N, C, dtype=torch.float32, device=U.device) for _ in range(2))
            ld = C
        # bytes: the CSR lists and positions, U / centre terms / five tables / gout once, dU and dVc written (rows k-fold from L2)
        with profiler.span("edge_bwd_tables", 4.0 * (3.0 * B * N * k + 10.0 * B * N * C)):
            ws = torch.empty(query("prifit_edge_bwd_workspace", B, N, k, C) // 8, dtype=torch.int64, device=U.device)
            call("prifit_edge_bwd", ptr(gout), _LL(gout.stride(0)), ptr(ystar), ptr(ysum), ptr(karg), ptr(scale), ptr(shift),
                 ptr(ca), ptr(cb), ptr(cd), ptr(U), _LL(U.shape[2]), ptr(vct), int(ctx.stacked), ptr(idx), ptr(offs), ptr(lst),
                 ptr(pos), B, N, k, C, _F(ctx.cfg["slope"]), ptr(dU), ptr(dVc), _LL(ld), ptr(ws), cur_stream())
        return dU, (None if ctx.stacked else dVc), None, None, dgamma, dbeta, None


def edge_csr(idx):
    """(offs [B,N+1], lst [B,N*k], pos [B,N*k]) int32: the in-edge lists of the neighbour graph idx [B,N,k] and where each
    edge sits in them (prifit_edge_csr)."""
    B, N, k = idx.shape
    offs = torch.empty(B, N + 1, dtype=torch.int32, device=idx.device)
    lst, pos = (torch.empty(B, N * k, dtype=torch.int32, device=idx.device) for _ in range(2))
    with profiler.span("edge_csr", 4.0 * (4.0 * B * N * k + B * N)):       # the lists read twice, lst and pos written
        call("prifit_edge_csr", ptr(idx), B, N, k, ptr(offs), ptr(lst), ptr(pos), cur_stream())
    return offs, lst, pos


# The edge convolution by linearity (default): W [x_j - x_i | x_i] = Wa x_j + (Wb - Wa) x_i = U_j - Vc_i with U = X Wa^T and
# Vc = X (Wa - Wb)^T computed once per POINT (two products over B N rows); per EDGE only a gather of Cout-wide rows of U
# (nn_ops.GatherLinearFn, the kernel of the set-abstraction first layers).  The [B N k, 2C] edge rows of upstream
# (src/dgcnn.py:98-105), the products over B N k rows (49 GFLOP forward at B = 24, k = 20) and their autograd (a dA and a dW
# product over the edge rows, a 2C-wide scatter) never exist.  PRIFIT_EDGE_LINEARITY=0: rows + product (A/B arm; tested).
_EDGE_LINEARITY = __import__("os").environ.get("PRIFIT_EDGE_LINEARITY", "1") != "0"
# ... with its pooled GroupNorm backward formed inside the scatter (0: apply pass writes dY, then the scatter; A/B arm, tested)
_EDGE_FUSED_BWD = __import__("os").environ.get("PRIFIT_EDGE_FUSED_BWD", "1") != "0"
# ... and, by default, with no per-edge tensor at all (EdgeConvTabFn; 0: the pre-activations are written and re-read; A/B arm, tested)
_EDGE_TABLES = __import__("os").environ.get("PRIFIT_EDGE_TABLES", "1") != "0"
# the global max over the cloud fused into the mlp1 block (0: activation pass + torch max; A/B arm, tested)
_GLOBAL_POOL_FUSED = __import__("os").environ.get("PRIFIT_GLOBAL_POOL_FUSED", "1") != "0"
# ... and its backward in the algebraic form: no [B N, Cout] tensor dY, the two products over it replaced by per-sample
# [Cin, Cin] products and B x Cout winners' rows (0: pool_bwd_apply + the dense dA / dW products; A/B arm, tested)
_GLOBAL_POOL_ALG = __import__("os").environ.get("PRIFIT_GLOBAL_POOL_ALG", "1") != "0"
# ... whose forward then does not store the product at all (0: stores it; A/B arm, tested)
_GLOBAL_POOL_NOSTORE = __import__("os").environ.get("PRIFIT_GLOBAL_POOL_NOSTORE", "1") != "0"


def _w2d(conv, kp=None):
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    if kp is not None and kp > w.shape[1]:
        w = torch.cat([w, w.new_zeros(w.shape[0], kp - w.shape[1])], dim=1)
    return w


class DGCNNEncoderGn(nn.Module):
    def __init__(self, input_channels=3, nn_nb=80, dilation=1):
        super().__init__()
        if input_channels not in (3, 6):
            raise ValueError("input_channels: 3 (xyz) or 6 (xyz + normals), as upstream src/dgcnn.py:171,199")
        self.k = nn_nb
        self.dilation_factor = dilation
        self.drop = 0.0
        self.input_channels = input_channels
        self.bn1 = nn.GroupNorm(2, 64)
        self.bn2 = nn.GroupNorm(2, 64)
        self.bn3 = nn.GroupNorm(2, 128)
        self.conv1 = nn.Sequential(nn.Conv2d(input_channels * 2, 64, kernel_size=1, bias=False), self.bn1,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.conv2 = nn.Sequential(nn.Conv2d(64 * 2, 64, kernel_size=1, bias=False), self.bn2,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.conv3 = nn.Sequential(nn.Conv2d(64 * 2, 128, kernel_size=1, bias=False), self.bn3,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.mlp1 = nn.Conv1d(256, 1024, 1)
        self.bnmlp1 = nn.GroupNorm(8, 1024)

    def _edge_conv(self, feats, idx, seq, N, csr=None):
        conv, gn = seq[0], seq[1]
        C = feats.shape[-1]
        k = idx.shape[2]
        cfg = {"groups": gn.num_groups, "rps": N * k, "slope": 0.2, "pool_K": k, "eps": gn.eps}
        B = feats.shape[0]
        Cout = conv.weight.shape[0]
        if _EDGE_LINEARITY and conv.bias is None and (N * k) % query("prifit_reduce_rows_per_slab") == 0 and Cout % 4 == 0:
            X = feats.reshape(B * N, C)
            pad = _pad4(C) - C                          # 16-byte rows for the product kernels (the 3 input coordinates)
            # (the backward's reduction runs over the [B N, Cout] table of winners in 128-row slabs: whole slabs per sample)
            if csr is not None and query("prifit_edge_tables_supported", N, k, Cout) and N % query("prifit_reduce_rows_per_slab") == 0:
                # ONE product X [Wa; Wb]^T = [U | Vb] per point (y = U_j - U_i + Vb_i) and one [B N, 2 Cout] gradient back;
                # [Wa; Wb] is a permuted copy of the weight (one launch each way, no slices to re-assemble in the backward)
                Wst = conv.weight.reshape(Cout, 2, C).permute(1, 0, 2).reshape(2 * Cout, C)
                if pad:
                    X, Wst = F.pad(X, (0, pad)), F.pad(Wst, (0, pad))
                UV = LinearFn.apply(X, Wst, None).view(B, N, 2 * Cout)
                return EdgeConvTabFn.apply(UV, None, idx, csr, gn.weight, gn.bias, cfg)   # [B*N, Cout]
            w = conv.weight.reshape(Cout, 2 * C)
            wa, wb = w[:, :C], w[:, C:]
            if pad:
                X = torch.cat([X, X.new_zeros(B * N, pad)], dim=1)
                wa = torch.cat([wa, wa.new_zeros(Cout, pad)], dim=1)
                wb = torch.cat([wb, wb.new_zeros(Cout, pad)], dim=1)
            U = LinearFn.apply(X, wa, None).view(B, N, Cout)             # neighbour term, per point
            Vc = LinearFn.apply(X, wa - wb, None).view(B, N, Cout)       # minus the centre term, per point
            if _EDGE_FUSED_BWD:
                return EdgeConvLinFn.apply(U, Vc, idx, gn.weight, gn.bias, cfg)      # [B*N, Cout]
            Y, slab = nn_ops.GatherLinearFn.apply(U, Vc, None, idx, True)
            return GNActFn.apply(Y, slab, gn.weight, gn.bias, cfg)
        ld = _pad4(2 * C)
        rows = EdgeGatherFn.apply(feats, idx, ld)
        return ConvGNActFn.apply(rows, _w2d(conv, ld), None, gn.weight, gn.bias, cfg)   # [B*N, Cout]

    def forward_cl(self, pts):
        """pts [B,N,3 or 6] -> (x4 [B,1024], x_features [B*N,256]) channels-last.  input_channels == 6 (upstream :199-222):
        the first graph comes from the normal-weighted metric and no layer applies the dilation factor."""
        B, N, _ = pts.shape
        normals = self.input_channels == 6
        k, k2 = self.k, self.k * (1 if normals else self.dilation_factor)
        step = k2 // k
        tables = _EDGE_TABLES and _EDGE_LINEARITY and _EDGE_FUSED_BWD
        with torch.no_grad():
            idx1 = (_knn_normals_cl(pts, k2) if normals else _knn_cl(pts, k2))[:, :, ::step].contiguous()
            csr1 = edge_csr(idx1) if tables and N <= 8192 else None
        x1 = self._edge_conv(pts, idx1, self.conv1, N, csr1)
        with torch.no_grad():
            idx2 = _knn_cl(x1.detach().view(B, N, -1), k2)[:, :, ::step].contiguous()
            csr2 = edge_csr(idx2) if tables and N <= 8192 else None
        x2 = self._edge_conv(x1.view(B, N, -1), idx2, self.conv2, N, csr2)
        x3 = self._edge_conv(x2.view(B, N, -1), idx2, self.conv3, N, csr2)      # re-uses the second graph (:191)
        feats = torch.cat((x1, x2, x3), dim=1)
        cfg = {"groups": self.bnmlp1.num_groups, "rps": N, "slope": 0.0, "pool_K": 0, "eps": self.bnmlp1.eps}
        if _GLOBAL_POOL_FUSED and N % 32 == 0 and pool_product_ok(B * N, self.mlp1.weight.shape[0], feats.shape[1]):
            # relu(gn(mlp1(.))) and the max over the cloud (upstream :194-197) as ONE pooled block with K = N: the [B N, 1024]
            # activation is neither written nor reduced by a separate pass, and the backward routes the gradient through the
            # winners' indices instead of a dense [B, N, 1024] tensor of mostly zeros
            cfg["pool_K"] = N
            x4 = ConvGNActFn.apply(feats, _w2d(self.mlp1), self.mlp1.bias, self.bnmlp1.weight, self.bnmlp1.bias, cfg)
            return x4, feats
        h = ConvGNActFn.apply(feats, _w2d(self.mlp1), self.mlp1.bias, self.bnmlp1.weight, self.bnmlp1.bias, cfg)
        x4 = h.view(B, N, -1).max(dim=1)[0]
        return x4, feats

    def forward(self, x):
        """x [B,3 or 6,N] -> (x4 [B,1024], x_features [B,256,N]) as upstream :171-222."""
        B, _, N = x.shape
        x4, feats = self.forward_cl(x.transpose(1, 2).contiguous())
        return x4, feats.view(B, N, -1).permute(0, 2, 1)


class DGCNGn(nn.Module):
    def __init__(self, emb_size=128, num_channels=3, nn_nb=80, dilation=1):
        super().__init__()
        self.encoder = DGCNNEncoderGn(input_channels=num_channels, nn_nb=nn_nb, dilation=dilation)
        self.drop = 0.0
        self.conv1 = nn.Conv1d(1024 + 256, 512, 1)
        self.bn1 = nn.GroupNorm(8, 512)
        self.conv2 = nn.Conv1d(512, 256, 1)
        self.bn2 = nn.GroupNorm(4, 256)
        self.emb_size = emb_size
        self.mlp_seg_prob1 = nn.Conv1d(256, 256, 1)
        self.mlp_seg_prob2 = nn.Conv1d(256, self.emb_size, 1, bias=False)
        self.bn_seg_prob1 = nn.GroupNorm(4, 256)
        self.mlp_segmentation = nn.Conv1d(256, 3, 1)

    def _block(self, x, conv, gn, N):
        cfg = {"groups": gn.num_groups, "rps": N, "slope": 0.0, "pool_K": 0, "eps": gn.eps}
        return ConvGNActFn.apply(x, _w2d(conv), conv.bias, gn.weight, gn.bias, cfg)

    def forward(self, points):
        B, _, N = points.shape
        x4, feats = self.encoder.forward_cl(points.transpose(1, 2).contiguous())
        if _EDGE_LINEARITY and query("prifit_gn_finalize_supported", self.conv1.weight.shape[0], self.bn1.num_groups):
            # upstream :253-257 repeats the global feature x4 [B,1024] over the N points and concatenates it with the 256
            # point features in front of conv1 (1280 -> 512).  The x4 part of that product is the same for every point of a
            # sample: ONE row per sample (24 x 1024 x 512) instead of N (49152 x 1024 x 512, 4/5 of the layer's flops, and the
            # [B N, 1280] input never exists); it enters the GroupNorm as a per-sample offset of the pre-activation.
            w = self.conv1.weight.reshape(self.conv1.weight.shape[0], -1)
            w4, wf = w[:, :1024].contiguous(), w[:, 1024:].contiguous()
            off = LinearFn.apply(x4.contiguous(), w4, self.conv1.bias)                      # [B, 512], bias included
            cfg = {"groups": self.bn1.num_groups, "rps": N, "slope": 0.0, "pool_K": 0, "eps": self.bn1.eps}
            x = ConvGNActFn.apply(feats, wf, None, self.bn1.weight, self.bn1.bias, cfg, off)
        else:
            x = torch.cat([x4.unsqueeze(1).expand(B, N, 1024).reshape(B * N, 1024), feats], dim=1)
            x = self._block(x, self.conv1, self.bn1, N)
        x_all = self._block(x, self.conv2, self.bn2, N)
        x = self._block(x_all, self.mlp_seg_prob1, self.bn_seg_prob1, N)
        seg = LinearFn.apply(x, _w2d(self.mlp_segmentation), self.mlp_segmentation.bias)
        emb = LinearFn.apply(x, _w2d(self.mlp_seg_prob2), None)
        return emb.view(B, N, -1), seg.view(B, N, 3).permute(0, 2, 1)


class get_model(nn.Module):
    """Adapter with the part-seg call surface the trainer expects for `'dgcnn' in args.model`
    (train_partseg_shapenet.py:226-228): DGCNN embedding (src/dgcnn.py) + convex loss.  The reference ships
    no such module; outputs follow the 5-tuple of the MSG model with `seg` from DGCNGn's head."""

    def __init__(self, num_part, normal_channel=False, k=20):
        super().__init__()
        self.net = DGCNGn(emb_size=128, num_channels=6 if normal_channel else 3, nn_nb=k)
        self.beta = 1

    def forward(self, xyz, cls_label=None, chamfer_points=0, include_convex_loss=False, quantile=0.01,
                msc_iterations=5, max_num_clusters=25, fit_inputs=None, **_unused):
        if xyz.is_cuda:
            from .. import arena as zero_pool
            zero_pool.begin_step(xyz.device)
        emb, seg = self.net(xyz)
        xyz = xyz[:, :3]                     # the loss sees positions only (normal_channel=True: rows 3..5 are normals)
        total = torch.zeros(1, device=xyz.device)
        chamfer = torch.zeros(1, device=xyz.device)
        extra = ()
        if include_convex_loss:
            from ..convex_loss import convex_loss
            if self.beta > 0.001:
                self.beta *= 0.99
            fe = emb.permute(0, 2, 1)
            total, chamfer, params, labels = convex_loss(xyz, chamfer_points, fe, quantile=quantile,
                                                         iterations=msc_iterations,
                                                         max_num_clusters=max_num_clusters, **(fit_inputs or {}))
            extra = (labels, params, fe)
        return (F.log_softmax(seg, dim=1).permute(0, 2, 1), None, emb.permute(0, 2, 1), total, chamfer) + extra
