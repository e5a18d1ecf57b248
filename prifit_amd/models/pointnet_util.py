"""MI355X backend behind the call surface of the reference's models/pointnet_util.py.

Same function / class names, constructor arguments, tensor layouts ([B, C, N] in and out) and
state_dict keys as upstream (models/pointnet_util.py:19-314); every op runs as a hand-written HIP
kernel through libprifit_hip.so.  There is no CPU fallback: tensors must live on the GPU.

Differences that are deliberate and documented (SURVEY.md section 8a'):
  * farthest_point_sample / the modules take an optional explicit FPS start index (`fps_start`);
    when omitted a random start is drawn exactly like upstream (:75);
  * internally activations are channels-last and input channels are zero-padded to multiples of 4
    (weights are re-packed on the fly; gradients flow back to the upstream-shaped parameters).
"""
import contextlib
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import arena as zero_pool
from .._lib import call, cur_stream, ptr, query
from ..nn_ops import (ConcatWindowsFn, FpRowsFn, GatherLinearFn, GroupGatherFn, LinearFn, SAGroupDirectFn, SAGroupGatherFn, SharedMLPFn,
                      ThreeInterpolateFn, _sa_group_launch, sa_group_supported)


# ------------------------------------------------------------------ functional surface (:19-107)

def _z(like, *shape):
    """Zero padding blocks from the step's one pre-zeroed pool (prifit_amd/arena.py) instead of a fill launch each."""
    if like.is_cuda and like.dtype == torch.float32:
        return zero_pool.zeros(*shape, device=like.device)
    return like.new_zeros(*shape)

def square_distance(src, dst):
    """upstream :19-40 -- src [B,N,3], dst [B,M,3] -> [B,N,M] (expanded form, bitwise)."""
    return ops.square_distance(src, dst)


def index_points(points, idx):
    """upstream :43-60 -- points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,(K,)C]."""
    B = points.shape[0]
    flat = idx.reshape(B, -1).long()
    out = torch.gather(points, 1, flat.unsqueeze(-1).expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def farthest_point_sample(xyz, npoint, start_idx=None):
    """upstream :63-84 -- xyz [B,N,3] -> int64 [B,npoint]."""
    return ops.farthest_point_sample(xyz, npoint, start_idx)


def query_ball_point(radius, nsample, xyz, new_xyz):
    """upstream :87-107 -- int64 [B,S,nsample]."""
    return ops.ball_query_multi([radius], [nsample], xyz, new_xyz, idx64=True)[0]


def _pad4(c):
    return (c + 3) // 4 * 4


_col_index_cache = {}


def _col_index(cols, ncols, device):
    """(map, inv_off, inv_idx) int32 device tensors of a column map `cols` (tuple; entry j = source column of output column j,
    -1 = zero column), cached per device: the map itself and the CSR of its inverse (for every source column the output
    columns it feeds, ascending)."""
    key = (cols, ncols, str(device))
    hit = _col_index_cache.get(key)
    if hit is None:
        inv = [[] for _ in range(ncols)]
        for j, c in enumerate(cols):
            if c >= 0:
                inv[c].append(j)
        off = [0]
        for lst in inv:
            off.append(off[-1] + len(lst))
        flat = [j for lst in inv for j in lst] or [0]
        hit = _col_index_cache[key] = tuple(torch.tensor(t, dtype=torch.int32, device=device) for t in (list(cols), off, flat))
    return hit


class PackColsFn(torch.autograd.Function):
    """out[:, j] = w[:, cols[j]] (cols[j] < 0: a zero column): a column permutation / padding of a weight matrix as one
    launch forward (prifit_pack_cols) and one backward (prifit_unpack_cols: a source column may feed several output columns --
    their gradients are summed).  Slices + cat do the same with a zero-fill and a copy per slice in the backward, index_select
    / index_add_ with 2 + 3 launches per weight: ~30 tiny launches per training step over the first layers."""

    @staticmethod
    def forward(ctx, w, cols):
        w = w.contiguous()
        cmap, inv_off, inv_idx = _col_index(cols, w.shape[1], w.device)
        ctx.idx, ctx.ncols = (inv_off, inv_idx), w.shape[1]
        out = torch.empty(w.shape[0], len(cols), dtype=torch.float32, device=w.device)
        call("prifit_pack_cols", ptr(w), w.shape[0], w.shape[1], ptr(cmap), len(cols), ptr(out), cur_stream())
        return out

    @staticmethod
    def backward(ctx, g):
        inv_off, inv_idx = ctx.idx
        g = g.contiguous()
        gw = torch.empty(g.shape[0], ctx.ncols, dtype=torch.float32, device=g.device)
        call("prifit_unpack_cols", ptr(g), g.shape[0], g.shape[1], ptr(inv_off), ptr(inv_idx), ctx.ncols, ptr(gw), cur_stream())
        return gw, None


class PackAllFn(torch.autograd.Function):
    """PackColsFn for every packed weight of a network at once: ONE launch forward (prifit_pack_cols_multi), one backward.
    apply(specs, *ws) -> tuple of packed matrices; specs[i] = the column map of ws[i]."""

    @staticmethod
    def forward(ctx, specs, *ws):
        import ctypes
        from ..nn_ops import _ptr_array
        ws = [w.contiguous() for w in ws]
        idx = [_col_index(cols, w.shape[1], w.device) for cols, w in zip(specs, ws)]
        outs = [torch.empty(w.shape[0], len(cols), dtype=torch.float32, device=w.device) for cols, w in zip(specs, ws)]
        n = len(ws)
        ints = lambda v: (ctypes.c_int32 * n)(*v)
        call("prifit_pack_cols_multi", n, _ptr_array(ws), ints([w.shape[0] for w in ws]), ints([w.shape[1] for w in ws]),
             _ptr_array([i[0] for i in idx]), ints([len(c) for c in specs]), _ptr_array(outs), cur_stream())
        ctx.idx, ctx.shapes = idx, [tuple(w.shape) for w in ws]
        ctx.set_materialize_grads(False)      # an output nobody used: None in the backward, its job is skipped
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        import ctypes
        from ..nn_ops import _ptr_array
        live = [i for i, g in enumerate(gs) if g is not None]
        gws = [None] * len(gs)
        if live:
            g = [gs[i].contiguous() for i in live]
            out = [torch.empty(ctx.shapes[i], dtype=torch.float32, device=g[0].device) for i in live]
            n = len(live)
            ints = lambda v: (ctypes.c_int32 * n)(*v)
            call("prifit_unpack_cols_multi", n, _ptr_array(g), ints([t.shape[0] for t in g]), ints([t.shape[1] for t in g]),
                 _ptr_array([ctx.idx[i][1] for i in live]), _ptr_array([ctx.idx[i][2] for i in live]),
                 ints([ctx.shapes[i][1] for i in live]), _ptr_array(out), cur_stream())
            for i, o in zip(live, out):
                gws[i] = o
        return (None, *gws)


class PackPlan:
    """The packed weights a network asks for during a forward, remembered (first forward) and then produced together at the
    top of every later forward (`with pack_plan(net):`): 8 + 8 launches of a training step become 1 + 1.  A site that is not in
    the plan -- the first forward, a weight that appeared later -- takes the single launch and joins."""

    def __init__(self):
        self.sites, self.keys, self.results = [], set(), None

    def begin(self):
        self.results = {}
        if self.sites and _PACK_ALL:
            outs = PackAllFn.apply(tuple(c for _, c in self.sites), *[p.reshape(p.shape[0], -1) for p, _ in self.sites])
            self.results = {(id(p), c): o for (p, c), o in zip(self.sites, outs)}

    def lookup(self, w, cols):
        base = w._base if w._base is not None else w
        key = (id(base), cols)
        hit = self.results.get(key)
        if hit is not None:
            return hit
        if key not in self.keys and isinstance(base, nn.Parameter) and len(self.sites) < _pack_max_jobs():
            self.sites.append((base, cols))
            self.keys.add(key)
        return None


_PACK_ALL = os.environ.get("PRIFIT_PACK_ALL", "1") != "0"    # 0: one launch per packed weight (A/B)
_plans = weakref.WeakKeyDictionary()
_active_plan = None


def _pack_max_jobs():
    return query("prifit_pack_cols_max_jobs")


@contextlib.contextmanager
def pack_plan(net):
    """Inside: `_pack_cols` serves the weights of `net` from one launch (see PackPlan)."""
    global _active_plan
    plan = _plans.get(net)
    if plan is None:
        plan = _plans[net] = PackPlan()
    outer, _active_plan = _active_plan, plan
    plan.begin()
    try:
        yield plan
    finally:
        plan.results = None
        _active_plan = outer


def _pack_cols(w, cols):
    cols = tuple(cols)
    if cols == tuple(range(w.shape[1])):
        return w
    if _active_plan is not None and w.is_cuda:
        hit = _active_plan.lookup(w, cols)
        if hit is not None:
            return hit
    return PackColsFn.apply(w, cols)


def _pack_weight(conv, perm_slices, kp):
    """[Cout, Cin(,1,1)] -> [Cout, kp]: columns re-ordered to the internal row layout, zero padded."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    cols = [c for a, b in perm_slices for c in range(a, b)]
    return _pack_cols(w, cols + [-1] * (kp - len(cols)))


def _mlp_tensors(convs, bns, first_weight):
    ts = []
    for i, (conv, bn) in enumerate(zip(convs, bns)):
        w = first_weight if i == 0 else conv.weight.reshape(conv.weight.shape[0], -1)
        ts += [w, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    return ts


# The first layer of a set-abstraction MLP is computed by linearity (no grouped tensor) when the grouped row would
# be wider than the layer's output; PRIFIT_SA_LINEARITY=0 keeps the gather + GEMM form for A/B measurements.
_SA_LINEARITY = os.environ.get("PRIFIT_SA_LINEARITY", "1") != "0"


def _use_linearity(conv, kp):
    return _SA_LINEARITY and kp > conv.weight.shape[0]


def _linearity_rows(feats, xyz, new_xyz, kp):
    """The two row tables every radius of a level multiplies: rows [B N, kp] = [feat | xyz | 0-pad] per point and
    c4 [B S, 4] = [centre | 0] per centre (built once per level, not once per radius)."""
    B, N, _ = xyz.shape
    D = 0 if feats is None else feats.shape[-1]
    S = new_xyz.shape[1]
    parts = ([feats] if feats is not None else []) + [xyz]
    if kp > D + 3:
        parts.append(_z(xyz, B, N, kp - D - 3))
    rows = torch.cat(parts, dim=-1).reshape(B * N, kp)
    c4 = torch.cat([new_xyz, _z(new_xyz, B, S, 1)], dim=-1).reshape(B * S, 4)
    return rows, c4


def _linearity_operands(conv, feats, xyz, new_xyz, kp, feat_first, fold_bias=False, tables=None):
    """U [B,N,C1] = [feat | xyz] W1^T per POINT and Vc [B,S,C1] = c W1x^T per CENTRE (two small GEMMs):
    conv1([feat_j | xyz_j - c_g]) = U_j - Vc_g + bias.  fold_bias: the bias is added to U (then y = U_j - Vc_g).
    tables: `_linearity_rows(...)` of the level when several radii share them."""
    B, N, _ = xyz.shape
    D = 0 if feats is None else feats.shape[-1]
    S = new_xyz.shape[1]
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    C1 = w.shape[0]
    if feat_first:   # upstream MSG order [features, rel_xyz] (:247)
        fcols, xcols = list(range(D)), list(range(D, D + 3))
    else:            # upstream single-scale order [rel_xyz, features] (:131)
        fcols, xcols = list(range(3, 3 + D)), list(range(3))
    rows, c4 = tables if tables is not None else _linearity_rows(feats, xyz, new_xyz, kp)
    w_pt = _pack_cols(w, fcols + xcols + [-1] * (kp - D - 3))
    U = LinearFn.apply(rows, w_pt, conv.bias if fold_bias else None).reshape(B, N, C1)
    Vc = LinearFn.apply(c4, _pack_cols(w, xcols + [-1]), None).reshape(B, S, C1)
    return U, Vc


def _first_layer_by_linearity(conv, bn_training, feats, xyz, new_xyz, idx, kp, feat_first):
    """conv1 over the grouped [features | rel_xyz] rows without materialising them: project every POINT
    once (U), every CENTRE once (Vc), then gather C1-wide rows (GatherLinearFn).  Used when the grouped row
    would be wider than the layer's output."""
    U, Vc = _linearity_operands(conv, feats, xyz, new_xyz, kp, feat_first)
    return GatherLinearFn.apply(U, Vc, conv.bias, idx, bn_training)


# Ball query + grouping + the first conv of every per-radius MLP in ONE launch (csrc/sa_group.hip);
# PRIFIT_SA_FUSED=0 keeps the separate ball-query / gather / GEMM launches for A/B measurements.
_SA_FUSED = os.environ.get("PRIFIT_SA_FUSED", "1") != "0"
# direct mode in training: the first layer's BatchNorm backward is fused into its weight-gradient kernel (0: separate
# bn_relu_bwd_apply pass + SAGroupDirectFn autograd, A/B)
_DIRECT_FUSED_BWD = os.environ.get("PRIFIT_SA_DIRECT_FUSED_BWD", "1") != "0"


def _fused_mode(first_convs, feats, N, nsamples, kp):
    """None (separate launches), "direct" (narrow data inputs: weights in registers) or "gather" (by linearity)."""
    if not (_SA_FUSED and _SA_LINEARITY) or not sa_group_supported(N, nsamples, [c.weight.shape[0] for c in first_convs]):
        return None
    D = 0 if feats is None else feats.shape[-1]
    if D in (0, 3, 6) and (feats is None or not feats.requires_grad):
        return "direct"
    if all(_use_linearity(c, kp) for c in first_convs):
        return "gather"
    return None


# Direct-mode levels (SA1) in training: the 64-wide first-layer rows of a scale are NOT stored when every consumer can
# re-form them from the L2-resident per-point table U (first layer by linearity, bias folded in) -- the forward product of
# layer 2, the one-pass backward of layer 2 and the first conv's weight-gradient reduction (nn_ops.SharedMLPFn, cfg
# "norows").  SA1 at B = 24: 0.6 GB less written and 1.8 GB less read per step (2.4 GB of the step's ~36 GB of HBM traffic).
# MEASURED AND NOT THE DEFAULT: the grouping launches drop from 0.21 to 0.125 ms, but every consumer is SLOWER on
# rows gathered from L2 than on rows streamed from HBM (forward product +6 us, one-pass backward +26 us, weight-gradient
# reduction +22 us per scale): c2 10.09 -> 10.17 ms, c3 15.19 -> 15.30 ms on one box, alternating runs.  These kernels are
# not limited by HBM bytes alone.  PRIFIT_SA_NOROWS=1 turns it on (both arms are tested).
_SA_NOROWS = os.environ.get("PRIFIT_SA_NOROWS", "0") != "0"


def _norows_scales(mode, conv_blocks, training, B, S, nsamples):
    """Per scale: can its first-layer rows stay unstored?  (64-wide first layer, whole 64-row tiles per centre, >= 3 layers,
    the second layer on the streaming forward kernel and the one-pass backward kernel.)"""
    from .._lib import dll
    from .. import nn_ops
    if not (_SA_NOROWS and mode == "direct" and training and _DIRECT_FUSED_BWD and nn_ops._FUSE_RED and nn_ops._FUSE_BN_APPLY and
            nn_ops._STREAM):
        return [False] * len(conv_blocks)
    out = []
    for convs, K in zip(conv_blocks, nsamples):
        P = B * S * K
        C1 = convs[0].weight.shape[0]
        ok = (len(convs) >= 3 and C1 == 64 and K % 64 == 0 and P < (1 << 24) and convs[1].weight.shape[1] == 64)
        if ok:
            C2 = convs[1].weight.shape[0]
            ok = bool(query("prifit_gemm_stream_supported", 0, P, C2, 64)) and bool(query("prifit_gemm_stream_bwd_supported", P, C2, 64, 0))
        out.append(ok)
    return out


def nn_ops_point_tables(first_convs, feats, xyz, new_xyz, feat_first):
    """U_r [B,N,C_r] (bias folded in) and Vc_r [B,S,C_r] of every radius in one launch (prifit_sa_point_tables)."""
    import ctypes
    from .._lib import call, cur_stream, ptr
    from ..nn_ops import _ptr_array
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    D = 0 if feats is None else feats.shape[-1]
    R = len(first_convs)
    Ws = [c.weight.detach().reshape(c.weight.shape[0], -1).contiguous() for c in first_convs]
    bs = [None if c.bias is None else c.bias.detach().contiguous() for c in first_convs]
    Us = [torch.empty(B, N, w.shape[0], dtype=torch.float32, device=xyz.device) for w in Ws]
    Vcs = [torch.empty(B, S, w.shape[0], dtype=torch.float32, device=xyz.device) for w in Ws]
    wd = (ctypes.c_int * R)(*[int(w.shape[0]) for w in Ws])
    call("prifit_sa_point_tables", ptr(xyz.contiguous()), ptr(new_xyz.contiguous()), ptr(None if feats is None else feats.contiguous()),
         B, N, S, D, int(feat_first), R, wd, _ptr_array(Ws), _ptr_array(bs), _ptr_array(Us), _ptr_array(Vcs), cur_stream())
    return Us, Vcs


def _fused_first_layers_norows(first_convs, norows, feats, xyz, new_xyz, radii, nsamples, kp, feat_first, first_bns=None):
    """Direct-mode level in training with some scales' rows unstored: ONE launch in the by-linearity form for all scales
    (y = U_j - Vc_g, U = [feat | xyz] W1^T + b per point, Vc = c W1x^T per centre: the index lists and the BatchNorm
    statistics of every scale, rows only where `norows` is False).  The first conv's weight gradient stays the direct
    reduction dW1 = dY1^T [feat | rel] (SharedMLPFn owns it), so U / Vc are plain data for autograd."""
    D = 0 if feats is None else feats.shape[-1]
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    with torch.no_grad():
        Us, Vcs = nn_ops_point_tables(first_convs, feats, xyz, new_xyz, feat_first)
        Ys, slabs, idxs = _sa_group_launch(1, xyz, new_xyz, None, True, list(radii), list(nsamples),
                                           [u.shape[-1] for u in Us], None, Us, Vcs, [None] * len(first_convs),
                                           rows=[not nr for nr in norows], bn=_bn_args(first_bns))
    out = []
    for i in range(len(first_convs)):
        info = {"idx": idxs[i], "xyz": xyz, "new_xyz": new_xyz, "feat": feats, "feat_first": feat_first, "K": nsamples[i], "D": D}
        if norows[i]:
            info.update(norows=True, U=Us[i], Vc=Vcs[i], P=B * S * nsamples[i], N=N, S=S)
        out.append((Ys[i], slabs[i], info))
    return out


def _bn_args(first_bns):
    """(gamma, beta, running_mean, running_var, eps, momentum) of every scale's first BatchNorm: the front end's launch
    finalizes the statistics itself (nn_ops._sa_group_launch(bn=...))."""
    if first_bns is None:
        return None
    return [(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, 0.1 if bn.momentum is None else bn.momentum)
            for bn in first_bns]


def _fused_first_layers(mode, first_convs, training, feats, xyz, new_xyz, radii, nsamples, kp, feat_first,
                        fused_gather_bwd=False, first_bns=None):
    """-> per radius (Y1 [B*S*K, C1], column-statistics slab, info): `info` is None, or -- direct mode in training --
    the dict SharedMLPFn needs to take over the first conv's weight gradient (cfg["preact_direct"]): Y1 is then plain
    data for autograd and the BatchNorm backward of that layer is fused into the weight-gradient kernel."""
    if mode == "direct" and training and _DIRECT_FUSED_BWD:
        D = 0 if feats is None else feats.shape[-1]
        with torch.no_grad():
            Ws = [c.weight.reshape(c.weight.shape[0], -1).contiguous() for c in first_convs]
            bs = [None if c.bias is None else c.bias.contiguous() for c in first_convs]
            fx = feats is not None and D == 3 and feats.data_ptr() == xyz.data_ptr()
            Ys, slabs, idxs = _sa_group_launch(0, xyz, new_xyz, feats, feat_first, list(radii), list(nsamples),
                                               [w.shape[0] for w in Ws], Ws, None, None, bs, feat_xyz=fx, bn=_bn_args(first_bns))
        return [(Ys[i], slabs[i], {"idx": idxs[i], "xyz": xyz, "new_xyz": new_xyz, "feat": feats, "feat_first": feat_first,
                                   "K": nsamples[i], "D": D}) for i in range(len(first_convs))]
    if mode == "gather" and training and fused_gather_bwd:
        # first layer by linearity, training: U / Vc stay differentiable, the launch itself is data for autograd, and the
        # SharedMLPFn that consumes Y1 returns dU / dVc with its first BatchNorm backward folded in (cfg["preact_gather"])
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        tables = _linearity_rows(feats, xyz, new_xyz, kp)
        ops_ = [_linearity_operands(c, feats, xyz, new_xyz, kp, feat_first, tables=tables) for c in first_convs]
        with torch.no_grad():
            Us = [u.detach().contiguous() for u, _ in ops_]
            Vcs = [v.detach().contiguous() for _, v in ops_]
            bs = [None if c.bias is None else c.bias.detach().contiguous() for c in first_convs]
            Ys, slabs, idxs = _sa_group_launch(1, xyz, new_xyz, None, True, list(radii), list(nsamples),
                                               [u.shape[-1] for u in Us], None, Us, Vcs, bs, bn=_bn_args(first_bns))
        return [(Ys[i], slabs[i], {"gather": True, "idx": idxs[i], "U": ops_[i][0], "Vc": ops_[i][1], "B": B, "N": N, "S": S,
                                   "K": nsamples[i]}) for i in range(len(first_convs))]
    if mode == "direct":
        ts = []
        for c in first_convs:
            ts += [c.weight.reshape(c.weight.shape[0], -1), c.bias]
        out = SAGroupDirectFn.apply(xyz, new_xyz, feats, (list(radii), list(nsamples), feat_first, training), *ts)
    else:
        ts = []
        tables = _linearity_rows(feats, xyz, new_xyz, kp)
        for c in first_convs:
            U, Vc = _linearity_operands(c, feats, xyz, new_xyz, kp, feat_first, tables=tables)
            ts += [U, Vc, c.bias]
        out = SAGroupGatherFn.apply(xyz, new_xyz, (list(radii), list(nsamples), training), *ts)
    return [(out[2 * i], out[2 * i + 1], None) for i in range(len(first_convs))]


def _mlp_tensors_preact(convs, bns, info=None):
    """Layer 0 is already computed: its W / bias slots are None -- unless SharedMLPFn owns the gradients of the fused
    front end's inputs: direct mode (`info` without "gather") the first conv's weight in the W slot; gather mode U in the
    W slot, the conv's bias in the bias slot and Vc behind the layer tensors."""
    if info is not None and info.get("gather"):
        ts = _mlp_tensors(convs, bns, info["U"])
        return ts + [info["Vc"]]
    if info is not None:
        return _mlp_tensors(convs, bns, convs[0].weight.reshape(convs[0].weight.shape[0], -1))
    ts = _mlp_tensors(convs, bns, None)
    ts[1] = None
    return ts


def _preact_cfg(cfg, slab, info):
    cfg["preact_slab"] = slab
    if info is not None and info.get("gather"):
        cfg["preact_gather"] = info
    else:
        cfg["preact_direct"] = info
    return cfg


# gather mode in training: dU / dVc of the first layer come out of ONE kernel with that layer's BatchNorm + ReLU backward
# folded in and the scatter staged in LDS (0: bn_relu_bwd_apply pass + SAGroupGatherFn autograd with global atomics, A/B)
_GATHER_FUSED_BWD = os.environ.get("PRIFIT_SA_GATHER_FUSED_BWD", "1") != "0"


def _gather_bwd_ok(mode, convs_per_scale, N):
    from .._lib import dll
    return (mode == "gather" and _GATHER_FUSED_BWD and all(len(c) >= 2 for c in convs_per_scale) and
            all(query("prifit_gather_linear_bwd_bn_supported", N, c[0].weight.shape[0]) for c in convs_per_scale))


_pending_counters = None   # list while a model forward batches the BatchNorm step counters, else None


class batched_bn_counters:
    """`num_batches_tracked += 1` of every BatchNorm touched inside the block as ONE multi-tensor add at exit
    (25 single-element kernels per forward otherwise)."""

    def __enter__(self):
        global _pending_counters
        self.prev, _pending_counters = _pending_counters, []
        return self

    def __exit__(self, *exc):
        global _pending_counters
        todo, _pending_counters = _pending_counters, self.prev
        if todo and exc[0] is None:
            torch._foreach_add_(todo, 1)
        return False


def _mlp_cfg(bns, pool_K, training):
    for bn in bns:
        if training and bn.num_batches_tracked is not None:
            if _pending_counters is not None:
                _pending_counters.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1
    return {"pool_K": pool_K, "training": training, "eps": bns[0].eps,
            "momentum": [0.1 if bn.momentum is None else bn.momentum for bn in bns]}


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, fps_start=None):
    """upstream :110-137 (SSG order [rel_xyz, features]); returns the reference's 4-D tensors."""
    B, N, C = xyz.shape
    fps_idx, new_xyz = ops.farthest_point_sample(xyz, npoint, fps_start, return_xyz=True)
    idx = ops.ball_query_multi([radius], [nsample], xyz, new_xyz)[0]
    D = 0 if points is None else points.shape[-1]
    rows = GroupGatherFn.apply(None if points is None else points.contiguous(), xyz.contiguous(), new_xyz, idx, 1,
                               _pad4(D + 3))
    new_points = rows[:, :D + 3].reshape(B, npoint, nsample, D + 3)
    if returnfps:
        return new_xyz, new_points, index_points(xyz, idx), fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """upstream :140-157."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device, dtype=xyz.dtype)
    grouped = xyz.view(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


# ------------------------------------------------------------------ modules (:160-314)
class PointNetSetAbstraction(nn.Module):
    """upstream :160-201 (single-scale or group_all).  state_dict: mlp_convs.i / mlp_bns.i."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def forward_cl(self, xyz, feats, fps_start=None):
        """channels-last: xyz [B,N,3], feats [B,N,D] or None -> (new_xyz [B,S,3], out [B,S,C'])."""
        B, N, _ = xyz.shape
        D = 0 if feats is None else feats.shape[-1]
        kp = _pad4(D + 3)
        if self.group_all:
            S, K = 1, N
            new_xyz = torch.zeros(B, 1, 3, device=xyz.device, dtype=xyz.dtype)
            parts = [xyz] + ([feats] if feats is not None else [])  # :154 order [xyz, features]
            if kp > D + 3:
                parts.append(_z(xyz, B, N, kp - D - 3))
            rows = torch.cat(parts, dim=-1).reshape(B * N, kp)
            w0 = _pack_weight(self.mlp_convs[0], [(0, D + 3)], kp)
        else:
            S, K = self.npoint, self.nsample
            _, new_xyz = ops.farthest_point_sample(xyz, S, fps_start, return_xyz=True)
            mode = _fused_mode([self.mlp_convs[0]], feats, N, [K], kp)
            if mode is not None:
                (y1, slab, info), = _fused_first_layers(mode, [self.mlp_convs[0]], self.training, feats, xyz, new_xyz,
                                                        [self.radius], [K], kp, feat_first=False,
                                                        fused_gather_bwd=_gather_bwd_ok(mode, [self.mlp_convs], N),
                                                        first_bns=[self.mlp_bns[0]] if self.training else None)
                cfg = _preact_cfg(_mlp_cfg(self.mlp_bns, K, self.training), slab, info)
                out = SharedMLPFn.apply(y1, cfg, *_mlp_tensors_preact(self.mlp_convs, self.mlp_bns, info))
                return new_xyz, out.reshape(B, S, -1)
            idx = ops.ball_query_multi([self.radius], [K], xyz, new_xyz)[0]
            # rows = [features, rel_xyz, pad]; upstream order is [rel_xyz, features] (:131)
            if _use_linearity(self.mlp_convs[0], kp):
                y1, slab = _first_layer_by_linearity(self.mlp_convs[0], self.training, feats, xyz, new_xyz, idx, kp,
                                                     feat_first=False)
                cfg = _mlp_cfg(self.mlp_bns, K, self.training)
                cfg["preact_slab"] = slab
                out = SharedMLPFn.apply(y1, cfg, *_mlp_tensors_preact(self.mlp_convs, self.mlp_bns))
                return new_xyz, out.reshape(B, S, -1)
            rows = GroupGatherFn.apply(feats, xyz, new_xyz, idx, 0, kp)
            w0 = _pack_weight(self.mlp_convs[0], [(3, 3 + D), (0, 3)], kp)
        out = SharedMLPFn.apply(rows, _mlp_cfg(self.mlp_bns, K, self.training),
                                *_mlp_tensors(self.mlp_convs, self.mlp_bns, w0))
        return new_xyz, out.reshape(B, S, -1)

    def forward(self, xyz, points, fps_start=None):
        """xyz [B,3,N], points [B,D,N] -> (new_xyz [B,3,S], new_points [B,D',S])."""
        x = xyz.permute(0, 2, 1).contiguous()
        f = points.permute(0, 2, 1).contiguous() if points is not None else None
        nx, out = self.forward_cl(x, f, fps_start)
        return nx.permute(0, 2, 1), out.permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """upstream :204-261.  state_dict: conv_blocks.i.j / bn_blocks.i.j."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for i in range(len(mlp_list)):
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last = in_channel + 3
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward_cl(self, xyz, feats, fps_start=None):
        B, N, _ = xyz.shape
        S = self.npoint
        D = 0 if feats is None else feats.shape[-1]
        kp = _pad4(D + 3)
        _, new_xyz = ops.farthest_point_sample(xyz, S, fps_start, return_xyz=True)
        pooled = []
        firsts = [blk[0] for blk in self.conv_blocks]
        mode = _fused_mode(firsts, feats, N, self.nsample_list, kp) if len(firsts) <= 4 else None
        if mode is not None:
            norows = _norows_scales(mode, self.conv_blocks, self.training, B, S, self.nsample_list)
            if any(norows):
                ys = _fused_first_layers_norows(firsts, norows, feats, xyz, new_xyz, self.radius_list, self.nsample_list, kp,
                                                feat_first=True, first_bns=[b[0] for b in self.bn_blocks])
            else:
                ys = _fused_first_layers(mode, firsts, self.training, feats, xyz, new_xyz, self.radius_list,
                                         self.nsample_list, kp, feat_first=True,
                                         fused_gather_bwd=_gather_bwd_ok(mode, self.conv_blocks, N),
                                         first_bns=[b[0] for b in self.bn_blocks] if self.training else None)
            # the scales write their pooled columns side by side into the level's output (no torch.cat)
            widths = [blk[-1].weight.shape[0] for blk in self.conv_blocks]
            wide = torch.empty(B * S, sum(widths), dtype=torch.float32, device=xyz.device)
            c0 = 0
            for i, K in enumerate(self.nsample_list):
                cfg = _preact_cfg(_mlp_cfg(self.bn_blocks[i], K, self.training), ys[i][1], ys[i][2])
                cfg["pool_out"] = wide[:, c0:c0 + widths[i]]
                c0 += widths[i]
                pooled.append(SharedMLPFn.apply(ys[i][0], cfg, *_mlp_tensors_preact(self.conv_blocks[i], self.bn_blocks[i],
                                                                                    ys[i][2])))
            return new_xyz, ConcatWindowsFn.apply(wide, *pooled).reshape(B, S, -1)
        idxs = ops.ball_query_multi(self.radius_list, self.nsample_list, xyz, new_xyz)  # one pass, all radii
        for i, K in enumerate(self.nsample_list):
            if _use_linearity(self.conv_blocks[i][0], kp):
                y1, slab = _first_layer_by_linearity(self.conv_blocks[i][0], self.training, feats, xyz, new_xyz,
                                                     idxs[i], kp, feat_first=True)
                cfg = _mlp_cfg(self.bn_blocks[i], K, self.training)
                cfg["preact_slab"] = slab
                pooled.append(SharedMLPFn.apply(y1, cfg, *_mlp_tensors_preact(self.conv_blocks[i], self.bn_blocks[i])))
                continue
            rows = GroupGatherFn.apply(feats, xyz, new_xyz, idxs[i], 0, kp)  # [features, rel_xyz] (:247)
            w0 = _pack_weight(self.conv_blocks[i][0], [(0, D + 3)], kp)
            pooled.append(SharedMLPFn.apply(rows, _mlp_cfg(self.bn_blocks[i], K, self.training),
                                            *_mlp_tensors(self.conv_blocks[i], self.bn_blocks[i], w0)))
        return new_xyz, torch.cat(pooled, dim=-1).reshape(B, S, -1)

    def forward(self, xyz, points, fps_start=None):
        x = xyz.permute(0, 2, 1).contiguous()
        f = points.permute(0, 2, 1).contiguous() if points is not None else None
        nx, out = self.forward_cl(x, f, fps_start)
        return nx.permute(0, 2, 1), out.permute(0, 2, 1)


class PointNetFeaturePropagation(nn.Module):
    """upstream :264-314.  state_dict: mlp_convs.i / mlp_bns.i."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last = out_channel

    def forward_cl(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,N,D1] or None, points2 [B,S,D2] -> [B,N,D']."""
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        D2 = points2.shape[-1]
        D1 = 0 if points1 is None else points1.shape[-1]
        kp = _pad4(D1 + D2)
        # internal row layout [interpolated, points1, pad]; upstream concatenates [points1, interpolated] (:306)
        if S > 1 or points1 is not None:
            idx, w = ops.three_nn(xyz1, xyz2) if S > 1 else (None, None)
            rows = FpRowsFn.apply(points2, idx, w, points1, kp)          # interpolation + concatenation + padding: one launch
        else:
            if S == 1:
                interp = points2.expand(B, N, D2).reshape(B * N, D2)
            else:
                idx, w = ops.three_nn(xyz1, xyz2)
                interp = ThreeInterpolateFn.apply(points2, idx, w)
            parts = [interp]
            if points1 is not None:
                parts.append(points1.reshape(B * N, D1))
            if kp > D1 + D2:
                parts.append(_z(interp, B * N, kp - D1 - D2))
            rows = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
        if len(self.mlp_convs) == 0:
            return rows[:, :D1 + D2].reshape(B, N, -1)
        w0 = _pack_weight(self.mlp_convs[0], [(D1, D1 + D2), (0, D1)], kp)
        out = SharedMLPFn.apply(rows.contiguous(), _mlp_cfg(self.mlp_bns, 0, self.training),
                                *_mlp_tensors(self.mlp_convs, self.mlp_bns, w0))
        return out.reshape(B, N, -1)

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D,N], points2 [B,D,S] -> [B,D',N]."""
        p1 = points1.permute(0, 2, 1).contiguous() if points1 is not None else None
        out = self.forward_cl(xyz1.permute(0, 2, 1).contiguous(), xyz2.permute(0, 2, 1).contiguous(), p1,
                              points2.permute(0, 2, 1).contiguous())
        return out.permute(0, 2, 1)
