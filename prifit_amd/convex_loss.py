"""Self-supervised convexity loss on the MI355X backend: call surface of the reference's convex_loss.py.

`convex_loss(points, chamfer_points, X, ...)` keeps upstream's positional/keyword arguments
(convex_loss.py:27) and return tuple `(total.view(1,1), l.view(1,1), ellipse_params_batch, labels)`
(:103).  The whole path -- normalise, mean-shift clustering, soft membership, weighted ellipsoid fit,
surface sampling, ellipsoid SDF and the analytic chamfer distance -- runs batched on the GPU.

Explicit inputs that replace hidden randomness / absent third-party code:
  * `rand_table`  the U[0,1) 3x3 matrices of src/ellipsoid_fitting.py:38 ([3,3] shared or [B,KM,3,3]);
                  drawn with torch.rand when omitted, like upstream;
  * `canonical`   pin the SVD column signs (largest component positive);
  * the surface sampler is the build's deterministic Fibonacci (U,V) table (trimesh is not used);
  * `center_ids`  (gradient-parity harness only) which point represents each mode -- last-bit noise in upstream's nms,
                  yet the gradient enters through that point (fit_ops._pin_representatives);
  * `embedding_offset` (benchmark harness only) a per-point tensor added to the embedding: stands in for training, which
                  is what separates the parts of a shape in embedding space (synth.part_embedding_offset).
Optional terms (off in the README configuration): `include_entropy_loss` (upstream :59-62,209-225),
`include_intersect_loss` (:96-99,374-413 -- upstream's scatter_mean import is commented out, so this term is
parity-unpinned and restates the documented intent), `include_pruning` (:78-82: upstream computes the pruned
set but never uses it in the loss, so the loss value is unchanged here too; `prune_points` itself (:444-470) is below and
`return_info=True` hands the pruned set out).  `if_cuboid` (:72-76) reads the fitted (r, V, c) as boxes
with half-sides r: cuboid SDF (:473-502), area-proportional budget (src/ellipsoid_utils.py:186-193) and
src/sample_ellipsoid.py:65-96 evaluated on the build's deterministic box-surface table (fit.hip cuboid_unit).
"""
import torch
import torch.nn.functional as F

from . import fit_ops
from . import arena as zero_pool
from .nn_ops import EPI_CHORD, NN, NT, gemm
from ._lib import call, cur_stream, ptr


class EllipseParams:
    """list[B] of list[K_b'] of (r[3], V[3,3], center[3]) -- upstream's `ellipse_params_batch`
    (src/ellipsoid_fitting.py:104-117) -- materialised lazily from the fixed-capacity tensors."""

    def __init__(self, r, V, c, valid, count):
        self.r, self.V, self.c, self.valid, self.count = r, V, c, valid, count
        self._lists = None

    def _materialise(self):
        if self._lists is None:
            valid = self.valid.cpu()
            self._lists = [[(self.r[b, k], self.V[b, k], self.c[b, k]) for k in range(valid.shape[1]) if valid[b, k]]
                           for b in range(valid.shape[0])]
        return self._lists

    def __len__(self):
        return self.r.shape[0]

    def __getitem__(self, b):
        return self._materialise()[b]

    def __iter__(self):
        return iter(self._materialise())


def analytic_chamfer_distance(r, V, c, valid, targets, cuboid=False):
    """src/utils.py:384-426: per shape (mean_s |s - NN_target(s)|^2 + mean_t (min_k |sdf_k(t)|)^2) / 2,
    averaged over the shapes that have at least one ellipsoid; zeros(1) if none has.  cuboid=True: the
    primitives are boxes with half-sides r (:396-397, convex_loss.py:473-502, src/sample_ellipsoid.py:65-96)."""
    M = targets.shape[1]
    sdf_sum = fit_ops.SdfLossFn.apply(targets, r, V, c, valid, cuboid)
    d2_sum, total = fit_ops.SampleNNLossFn.apply(r, V, c, valid, targets, cuboid)
    loss, part = fit_ops.ChamferCombineFn.apply(d2_sum, total, sdf_sum, valid, M)
    return loss, (part[0], part[1])


class EntropyFn(torch.autograd.Function):
    """convex_loss.py:209-225: mean over shapes of sum((1 + X_b X_b^T)^2) / n^2 (before the margin/relu).
    X [B,n,D] unit rows.  T = 1 + X X^T comes from the chord GEMM (2 - 2s -> T = 2 - chord/2)."""

    @staticmethod
    def forward(ctx, X):
        X = X.contiguous()
        B, n, D = X.shape
        T = torch.empty(B, n, n, dtype=torch.float32, device=X.device)
        gemm(NT, n, n, D, X, D, X, D, T, n, batch=B, sA=n * D, sB=n * D, sC=n * n, epi=EPI_CHORD)
        T.mul_(-0.5).add_(2.0)
        ctx.save_for_backward(X, T)
        return (T * T).sum(dim=(1, 2)).mean() / (n * n)

    @staticmethod
    def backward(ctx, g):
        X, T = ctx.saved_tensors
        B, n, D = X.shape
        dX = torch.empty_like(X)  # d/dX_b sum (1+s)^2 = 4 (1+S) X_b (S symmetric)
        gemm(NN, n, D, n, T, n, X, D, dX, D, batch=B, sA=n * n, sB=n * D, sC=n * D)
        return dX * (4.0 * g / (B * n * n))


def entropy(X, margin=1.8):
    return torch.relu(EntropyFn.apply(X) - margin)


class SdfMatrixFn(torch.autograd.Function):
    """sdf [B,M,KM] of every live ellipsoid at every point (convex_loss.py:331-343), 0 in dead slots."""

    @staticmethod
    def forward(ctx, points, r, V, c, valid, cuboid=False):
        points, r, V, c = points.contiguous(), r.contiguous(), V.contiguous(), c.contiguous()
        B, M, _ = points.shape
        K = r.shape[1]
        out = torch.empty(B, M, K, dtype=torch.float32, device=points.device)
        ctx.prim = "cuboid" if cuboid else "ellipsoid"   # convex_loss.py:381-384
        call("prifit_%s_sdf_matrix_fwd" % ctx.prim, ptr(points), B, M, ptr(r), ptr(V), ptr(c), ptr(valid), K, ptr(out),
             cur_stream())
        ctx.save_for_backward(points, r, V, c, valid)
        return out

    @staticmethod
    def backward(ctx, g):
        points, r, V, c, valid = ctx.saved_tensors
        B, M, _ = points.shape
        K = r.shape[1]
        g_r, g_V, g_c = zero_pool.zeros_like(r), zero_pool.zeros_like(V), zero_pool.zeros_like(c)
        call("prifit_%s_sdf_matrix_bwd" % ctx.prim, ptr(points), B, M, ptr(r), ptr(V), ptr(c), ptr(valid),
             ptr(g.contiguous()), K, ptr(g_r), ptr(g_V), ptr(g_c), cur_stream())
        return None, g_r, g_V, g_c, None, None


def intersection_loss_volume_3(r, V, c, valid, points, cuboid=False):
    """convex_loss.py:374-413 (intent; see the module docstring): per point the mean, over the ellipsoids it does
    not belong to, of clamp_max(sdf, -1e-3), squared; mean over points and over shapes with > 1 ellipsoid."""
    sdf = torch.clamp_max(SdfMatrixFn.apply(points, r, V, c, valid, cuboid), -1e-3)
    live = (valid != 0).unsqueeze(1)                                     # [B,1,KM]
    nlive = live.sum(dim=2).to(sdf.dtype)                                # [B,1]
    own = sdf.masked_fill(~live, float("inf")).argmin(dim=2, keepdim=True)
    mask = live & (torch.arange(sdf.shape[2], device=sdf.device).view(1, 1, -1) != own)
    others = (sdf * mask).sum(dim=2) / (nlive - 1).clamp(min=1.0)
    per = (others ** 2).mean(dim=1)
    has = (nlive.squeeze(1) > 1).to(sdf.dtype)
    return (per * has).sum() / has.sum().clamp(min=1.0)


# ------------------------------------------------------------------------------------------------
# The reference's list-based names of the SDF / intersection / pruning helpers (convex_loss.py:313-343, :374-413,
# :444-502), thin adapters over the batched kernels above: a caller that imports them by name gets the same numbers.
# ------------------------------------------------------------------------------------------------
def _sdf_lists(points, params_batch, cuboid):
    """list[B] of list[K_b] of [n_b] SDF vectors.  points: [B,M,3] tensor or list of [n_b,3] tensors (ragged, as the
    resampled surface points are); a non-tensor entry (upstream's -1 marker, src/ellipsoid_utils.py:116) gives []."""
    from .src.utils import pack_params
    plist = list(points.unbind(0)) if torch.is_tensor(points) else list(points)
    tens = [p for p in plist if torch.is_tensor(p)]
    if not tens:
        return [[] for _ in plist]
    dev = tens[0].device
    r, V, c, valid = pack_params(params_batch, dev)
    M = max(p.shape[0] for p in tens)
    pad = torch.zeros(len(plist), M, 3, dtype=torch.float32, device=dev)
    for b, p in enumerate(plist):
        if torch.is_tensor(p):
            pad[b, :p.shape[0]] = p
    sdf = SdfMatrixFn.apply(pad, r, V, c, valid, cuboid)                 # [B, M, KM], dead slots 0
    live = valid.cpu()
    out = []
    for b, p in enumerate(plist):
        if not torch.is_tensor(p):
            out.append([])
            continue
        out.append([sdf[b, :p.shape[0], k] for k in range(live.shape[1]) if live[b, k]])
    return out


def compute_sdf_ellipsoid(points, center, r, V):
    """convex_loss.py:313-328: approximate SDF of one ellipsoid at points [M,3] -> [M]."""
    return _sdf_lists(points.unsqueeze(0), [[(r, V, center)]], False)[0][0]


def compute_sdf_ellipsoids(points, ellipsoids_parameters):
    """:331-336: list over the ellipsoids (r, V, center) of one shape."""
    return _sdf_lists(points.unsqueeze(0), [ellipsoids_parameters], False)[0]


def compute_sdf_ellipsoids_batch(points, ellipsoids_parameters_batch):
    """:339-343: list[B] of list[K_b] of [M]."""
    return _sdf_lists(points, ellipsoids_parameters_batch, False)


def compute_sdf_cuboid(points, center, r, V):
    """:473-488: SDF of a box with half-sides r."""
    return _sdf_lists(points.unsqueeze(0), [[(r, V, center)]], True)[0][0]


def compute_sdf_cuboids(points, ellipsoids_parameters):
    """:491-496"""
    return _sdf_lists(points.unsqueeze(0), [ellipsoids_parameters], True)[0]


def compute_sdf_cuboid_batch(points, ellipsoids_parameters_batch):
    """:499-502"""
    return _sdf_lists(points, ellipsoids_parameters_batch, True)


def compute_intersection_loss_volume_3(ellipsoid_params_batch, points, cuboid=False):
    """:374-413 by its upstream name and argument order (see intersection_loss_volume_3)."""
    from .src.utils import pack_params
    r, V, c, valid = pack_params(ellipsoid_params_batch, points.device)
    return intersection_loss_volume_3(r, V, c, valid, points.contiguous(), cuboid=cuboid)


# The intersection-loss variants upstream keeps but never calls (only volume_3 is, convex_loss.py:98).  They are here so
# that code importing them by name keeps working; the two that evaluate the standard SDF at given points run on the HIP
# SDF kernel, the other three use formulas of their own (or need the gradient through the sampled points) and are torch
# compositions on the device.
def _zero_loss(device):
    return torch.zeros(1, device=device).requires_grad_(True)             # convex_loss.py:158, :275, :369


def _local_sdf(points, params, kind):
    """[n,K] of one shape in torch: 'quotient' = k0 (k0 - 1) / k1 without the +1e-6 of compute_sdf_ellipsoid (:132-135),
    'ellipsoid' = with it (:324-327), 'boxmax' = max_i(|q_i| - r_i) (:181-184)."""
    r = torch.stack([p[0] for p in params])                              # [K,3]
    V = torch.stack([p[1] for p in params])                              # [K,3,3]
    c = torch.stack([p[2] for p in params])                              # [K,3]
    q = torch.einsum("kji,nkj->nki", V, points[:, None, :] - c[None])    # V^T (p - c)
    if kind == "boxmax":
        return (q.abs() - r[None]).max(dim=2)[0]
    k0 = torch.norm(q / (r[None] + 1e-6), p=2, dim=2)
    k1 = torch.norm(q / (r[None] ** 2 + 1e-6), p=2, dim=2)
    return k0 * (k0 - 1.0) / (k1 + 1e-6 if kind == "ellipsoid" else k1)


def _surface_intersection(params_batch, sampled_points_batch, n, kind):
    per = []
    for b in range(n):
        s = _local_sdf(sampled_points_batch[b], params_batch[b], kind).min(dim=1)[0]
        per.append(torch.clamp_max(s, -1e-3).mean())
    return (torch.stack(per) ** 2).mean()


def compute_intersection_loss(ellipsoid_params_batch, sampled_points_batch):
    """:106-160: per shape mean over its sampled surface points of clamp_max(min_k sdf_k, -1e-3), squared, mean over the
    len(sampled_points_batch) shapes; zeros(1) for an empty batch."""
    if len(sampled_points_batch) == 0:
        return _zero_loss(torch.device("cuda"))
    return _surface_intersection(ellipsoid_params_batch, sampled_points_batch, len(sampled_points_batch), "quotient")


def compute_intersection_loss_cuboid(ellipsoid_params_batch, sampled_points_batch):
    """:163-206: the same with the box distance max_i(|q_i| - r_i), over len(ellipsoid_params_batch) shapes."""
    return _surface_intersection(ellipsoid_params_batch, sampled_points_batch, len(ellipsoid_params_batch), "boxmax")


def sample_axis(r, V, center, num_samples=40):
    """:285-310: points on the principal axes at ratios linspace(-0.9, 0.897, n_i) of the half-lengths, n_i =
    int(r_i * num_samples / sum r) + 1.  The counts are read on the host (they size the result)."""
    axes = (V * r.view(1, 3)).t()
    with torch.no_grad():
        n = ((r * num_samples / torch.sum(r)).int() + 1).tolist()
    rows = [axes[i:i + 1] * torch.linspace(-0.9, 0.897, n[i]).view(-1, 1).to(r.device) for i in range(3)]
    return torch.cat(rows, 0) + center.view(1, 3)


def compute_intersection_loss_volume(ellipsoid_params_batch, sampled_points_batch):
    """:227-282 as WRITTEN: the loop over j != i evaluates ellipsoid i's axis samples against ellipsoid i itself (:252
    indexes [b][i]), K-1 identical rows; per ellipsoid mean(clamp_max(sdf, -1e-3)), per shape the mean of the squares,
    shapes with <= 1 ellipsoid skipped.  The samples carry gradient to (r, V, c), so this one is composed in torch."""
    if len(sampled_points_batch) == 0:
        return _zero_loss(torch.device("cuda"))
    losses = []
    for b in range(len(sampled_points_batch)):
        params = ellipsoid_params_batch[b]
        if len(params) <= 1:
            continue
        per = [torch.clamp_max(_local_sdf(sample_axis(r, V, c), [(r, V, c)], "ellipsoid")[:, 0], -1e-3).mean()
               for r, V, c in params]
        losses.append((torch.stack(per) ** 2).mean())
    if not losses:
        return _zero_loss(sampled_points_batch[0].device if torch.is_tensor(sampled_points_batch[0]) else torch.device("cuda"))
    return torch.stack(losses).mean()


def _clamped_sdf_matrices(ellipsoid_params_batch, points, skip):
    sdfs = _sdf_lists(points, ellipsoid_params_batch, False)              # HIP SDF kernel, all shapes in one launch
    return [torch.clamp_max(torch.stack(s, 1), -1e-3) for s in sdfs if not skip(len(s))]


def compute_intersection_loss_volume_2(ellipsoid_params_batch, points):
    """:346-371: clamp_max(sdf, -1e-3) minus its detached row minimum, squared, mean; shapes with <= 1 ellipsoid skipped."""
    mats = _clamped_sdf_matrices(ellipsoid_params_batch, points, lambda k: k <= 1)
    if not mats:
        return _zero_loss(points[0].device)
    return torch.stack([((m - m.min(dim=1, keepdim=True)[0].detach()) ** 2).mean() for m in mats]).mean()


def compute_intersection_loss_volume_4(ellipsoid_params_batch, points):
    """:416-441: per point sum_k clamp_max(sdf_k, -1e-3)^2 minus the squared row minimum, mean; shapes with exactly one
    ellipsoid skipped (:427); a shape with none is skipped too (upstream's torch.stack([]) would raise)."""
    mats = _clamped_sdf_matrices(ellipsoid_params_batch, points, lambda k: k <= 1)
    if not mats:
        return _zero_loss(points[0].device)
    return torch.stack([((m ** 2).sum(dim=1) - m.min(dim=1)[0] ** 2).mean() for m in mats]).mean()


def prune_points(points, ellipsoid_param_batch, thres=-1e-3):
    """:444-470: of the predicted surface points of every shape keep those whose SDF with respect to the UNION of the
    shape's ellipsoids (min over k) is above `thres`, i.e. drop points well inside another primitive.  points: list[B]
    of [n_b,3] (the resampled points) or a [B,M,3] tensor; returns list[B] of [n_b',3].  The mask is computed without
    gradient (upstream :459-468); the kept points keep theirs."""
    with torch.no_grad():
        sdfs = _sdf_lists([p.detach() if torch.is_tensor(p) else p for p in points], ellipsoid_param_batch, False)
    pruned = []
    for b in range(len(sdfs)):
        p = points[b]
        if not torch.is_tensor(p) or not sdfs[b]:
            pruned.append(p)
            continue
        with torch.no_grad():
            keep = torch.stack(sdfs[b], 1).min(dim=1)[0] > thres
        pruned.append(p[keep])
    return pruned


def convex_loss(points, chamfer_points, X, batch_id=0, epoch=-1, seed=0, N=500, quantile=0.01, iterations=5,
                visualize=False, max_num_clusters=25, class_list=[], include_intersect_loss=False, alpha=1, beta=1,
                if_cuboid=False, include_pruning=False, include_entropy_loss=False, evaluation=False,
                rand_table=None, canonical=True, return_info=False, entropy_indices=None, intersect_jitter=None,
                center_ids=None, embedding_offset=None):
    """points [B,3,N], chamfer_points [B,3,M], X [B,D,N] (per-point embedding).
    embedding_offset [B,N,D] (benchmark harness only, synth.part_embedding_offset): added to the embedding before the
    normalisation."""
    emb = X.permute(0, 2, 1)
    if embedding_offset is not None:
        emb = emb + embedding_offset
    if emb.shape[2] <= 256 and emb.dtype == torch.float32:
        emb = fit_ops.Normalize2Fn.apply(emb)                # normalised twice upstream (:41,57), one kernel here
    else:
        emb = F.normalize(emb, dim=2, p=2)
        emb = F.normalize(emb, dim=2, p=2).contiguous()
    pts = points.permute(0, 2, 1).contiguous()
    entropy_loss = None
    if include_entropy_loss:                              # :59-62: a random quarter of the points
        if entropy_indices is None:
            entropy_indices = torch.randperm(emb.shape[1], device=emb.device)[: emb.shape[1] // 4]
        entropy_loss = entropy(emb[:, entropy_indices.to(emb.device)])
    cl = fit_ops.cluster(emb, quantile, iterations, max_num_clusters, center_ids=center_ids)   # clustering(): :68
    if rand_table is None:
        rand_table = torch.rand(pts.shape[0], cl["W"].shape[2], 3, 3, device=pts.device)
    r, V, c, valid = fit_ops.EllipsoidFitFn.apply(pts, cl["W"], cl["count"], rand_table.to(pts.device), canonical)  # :70
    if evaluation is False:
        tgt = chamfer_points.permute(0, 2, 1).contiguous()
        l, parts = analytic_chamfer_distance(r, V, c, valid, tgt, cuboid=if_cuboid)   # :72-89
    else:
        l, parts = torch.zeros((), device=pts.device, requires_grad=True), None
    intersection_loss = None
    if include_intersect_loss and evaluation is False:                   # :96-99 (targets jittered by U[0, 0.2))
        if intersect_jitter is None:
            intersect_jitter = torch.rand_like(tgt) * 0.2
        intersection_loss = intersection_loss_volume_3(r, V, c, valid, tgt - intersect_jitter.to(tgt.device),
                                                       cuboid=if_cuboid)
    total = l                                                            # :101: l + alpha * intersection + beta * entropy
    if intersection_loss is not None:
        total = total + alpha * intersection_loss
    if entropy_loss is not None:
        total = total + beta * entropy_loss
    params = EllipseParams(r, V, c, valid, cl["count"])
    labels = list(cl["labels"].unbind(0))
    if return_info:
        info = {"cluster": cl, "parts": parts, "r": r, "V": V, "c": c, "valid": valid}
        if include_pruning:
            # :78-82: upstream builds the pruned set and then feeds the UNPRUNED points to the loss (:89); the loss above
            # does the same.  The set itself is handed out here for callers that want it (visualisation upstream).
            from .src import ellipsoid_utils as eu
            resampled = (eu.sample_from_pred_params_cuboid if if_cuboid else eu.sample_from_pred_params)(params, N)
            info["pruned_points"] = prune_points(resampled, params)
        return total.view(1, 1), l.view(1, 1), params, labels, info
    return total.view(1, 1), l.view(1, 1), params, labels
