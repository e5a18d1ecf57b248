"""Call surface of the reference's src/ellipsoid_fitting.py on the MI355X backend."""
import torch

from .. import fit_ops


def weighted_ellipsoid_fitting_batch(points, weights, batch_id=0, rand_table=None, canonical=True):
    """upstream :104-117.  points [B,N,3]; weights: list of B tensors [N, K_b] (K_b <= 32).
    Returns list[B] of list of (r[3], V[3,3], center[3]); ill-conditioned fits are dropped (:43-47)."""
    B, N, _ = points.shape
    km = fit_ops.slots_for(max([w.shape[1] for w in weights] + [1]))
    W = torch.zeros(B, N, km, device=points.device)
    count = torch.zeros(B, dtype=torch.int32, device=points.device)
    for b, w in enumerate(weights):
        W[b, :, :w.shape[1]] = w
        count[b] = w.shape[1]
    if rand_table is None:
        rand_table = torch.rand(B, km, 3, 3, device=points.device)
    r, V, c, valid = fit_ops.EllipsoidFitFn.apply(points, W, count, rand_table, canonical)
    valid = valid.cpu()
    return [[(r[b, k], V[b, k], c[b, k]) for k in range(int(count[b])) if valid[b, k]] for b in range(B)]


def weighted_ellipsoids_fitting(points, weights, batch_id=0, shape_id=0, rand_table=None, canonical=True):
    """upstream :74-102 (one shape)."""
    return weighted_ellipsoid_fitting_batch(points.unsqueeze(0), [weights], rand_table=rand_table, canonical=canonical)[0]


def weighted_ellipsoid_fitting(points, weights, batch_id=0, shape_id=0, cluster_id=0, rand_table=None, canonical=True):
    """upstream :19-69 for ONE cluster: points [N,3], weights [N,1] (or [N]) -> (r[3], V[3,3], center[3]), or -1 when the
    fit is rejected (covariance condition number above 1e5, :43-47)."""
    w = weights.reshape(points.shape[0], 1)
    fit = weighted_ellipsoid_fitting_batch(points.unsqueeze(0), [w], rand_table=rand_table, canonical=canonical)[0]
    return fit[0] if fit else -1


def principal_axis_ellipsoid(points, weights, S, V, mode="slow"):
    """upstream :119-141: axis lengths of one cluster given the SVD (S, V) of its covariance.  "fast":
    sqrt(clamp(S, 1e-7)) * 1.732; "slow" (what the fit uses): the weight-scaled, re-centred points projected on V
    (third column flipped when V is a reflection), half the extent per axis.  Plain device arithmetic -- inside the fit
    kernel (csrc/fit.hip) the same steps run fused; this stand-alone form serves callers that bring their own SVD."""
    if mode == "fast":
        return torch.sqrt(torch.clamp(S, min=1e-7)) * 1.732, V
    w = weights.reshape(points.shape[0], 1)
    pts = (points - torch.sum(points * w, 0) / torch.sum(w)) * w
    if torch.det(V.T) < 0:
        V = torch.stack([V[:, 0], V[:, 1], -1 * V[:, 2]], 1)
    t = pts @ V
    return torch.abs(t.max(dim=0)[0] - t.min(dim=0)[0]) / 2.0, V
