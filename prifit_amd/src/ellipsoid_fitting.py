"""Call surface of the reference's src/ellipsoid_fitting.py on the MI355X backend."""
import torch

from .. import fit_ops


def weighted_ellipsoid_fitting_batch(points, weights, batch_id=0, rand_table=None, canonical=True):
    """upstream :104-117.  points [B,N,3]; weights: list of B tensors [N, K_b] (K_b <= 32).
    Returns list[B] of list of (r[3], V[3,3], center[3]); ill-conditioned fits are dropped (:43-47)."""
    B, N, _ = points.shape
    W = torch.zeros(B, N, fit_ops.KM, device=points.device)
    count = torch.zeros(B, dtype=torch.int32, device=points.device)
    for b, w in enumerate(weights):
        W[b, :, :w.shape[1]] = w
        count[b] = w.shape[1]
    if rand_table is None:
        rand_table = torch.rand(B, fit_ops.KM, 3, 3, device=points.device)
    r, V, c, valid = fit_ops.EllipsoidFitFn.apply(points, W, count, rand_table, canonical)
    valid = valid.cpu()
    return [[(r[b, k], V[b, k], c[b, k]) for k in range(int(count[b])) if valid[b, k]] for b in range(B)]


def weighted_ellipsoids_fitting(points, weights, batch_id=0, shape_id=0, rand_table=None, canonical=True):
    """upstream :74-102 (one shape)."""
    return weighted_ellipsoid_fitting_batch(points.unsqueeze(0), [weights], rand_table=rand_table, canonical=canonical)[0]
