"""Call surface of the loss-side helpers of the reference's src/utils.py: chamfer_distance_kdtree (:361-381) and
analytic_chamfer_distance (:384-426).  The host KD-tree of upstream is replaced by an exact nearest-neighbour search
on the device (any exact search agrees up to ties); the SDF half runs in the HIP kernels of csrc/fit.hip.

The fused form of analytic_chamfer_distance used by the training step (sampling + search in one kernel, fixed-capacity
parameter tensors) is prifit_amd.convex_loss.analytic_chamfer_distance; this module keeps upstream's list-based
signature for stand-alone callers (fitting.py)."""
import torch

from .. import fit_ops


def nearest_index(src, tgt, chunk=4096):
    """index of the exact nearest row of tgt [T,3] for every row of src [S,3] (direct differences, no expansion)."""
    out = []
    with torch.no_grad():
        for i in range(0, src.shape[0], chunk):
            d = ((src[i:i + chunk, None, :] - tgt[None, :, :]) ** 2).sum(-1)
            out.append(d.argmin(dim=1))
    return torch.cat(out)


def chamfer_distance_kdtree(source_points, target_points, sqrt=False):
    """upstream :361-381: source_points [B,S,3], target_points [B,T,3] -> mean over shapes of
    (mean_t |t - NN_source(t)|^2 + mean_s |s - NN_target(s)|^2) / 2."""
    per = []
    for b in range(source_points.shape[0]):
        s, t = source_points[b], target_points[b]
        d_st = ((t - s[nearest_index(t, s)]) ** 2).sum(1)
        d_ts = ((s - t[nearest_index(s, t)]) ** 2).sum(1)
        if sqrt:
            d_st, d_ts = torch.sqrt(d_st), torch.sqrt(d_ts)
        per.append((d_st.mean() + d_ts.mean()) / 2.0)
    return torch.stack(per).mean()


def pack_params(ellipsoid_params_batch, device):
    """list[B] of list[K_b] of (r[3], V[3,3], c[3]) -> fixed-capacity (r, V, c, valid) tensors, differentiable."""
    from ..convex_loss import EllipseParams
    if isinstance(ellipsoid_params_batch, EllipseParams):
        p = ellipsoid_params_batch
        return p.r, p.V, p.c, p.valid
    B = len(ellipsoid_params_batch)
    KM = fit_ops.slots_for(max([len(prm) for prm in ellipsoid_params_batch] + [1]))
    rows_r, rows_V, rows_c = [], [], []
    valid = torch.zeros(B, KM, dtype=torch.int32, device=device)
    for b, prm in enumerate(ellipsoid_params_batch):
        valid[b, :len(prm)] = 1
        pad = KM - len(prm)
        rows_r.append(torch.stack([p[0] for p in prm] + [torch.ones(3, device=device)] * pad))
        rows_V.append(torch.stack([p[1] for p in prm] + [torch.eye(3, device=device)] * pad))
        rows_c.append(torch.stack([p[2] for p in prm] + [torch.zeros(3, device=device)] * pad))
    return torch.stack(rows_r), torch.stack(rows_V), torch.stack(rows_c), valid


def analytic_chamfer_distance(ellipsoid_params_batch, source_points, target_points, cuboid=False):
    """upstream :384-426: per shape (mean_s |s - NN_target(s)|^2 + mean_t (min_k |sdf_k(t)|)^2) / 2, averaged over the
    shapes that have source points; zeros(1) when none has (:421-423)."""
    dev = target_points.device
    r, V, c, valid = pack_params(ellipsoid_params_batch, dev)
    M = target_points.shape[1]
    sdf_ts = fit_ops.SdfLossFn.apply(target_points.contiguous(), r, V, c, valid, cuboid) / M     # [B]
    per = []
    for b in range(target_points.shape[0]):
        s = source_points[b]
        if not torch.is_tensor(s):
            continue
        d_st = ((s - target_points[b][nearest_index(s, target_points[b])]) ** 2).sum(1)
        per.append((d_st.mean() + sdf_ts[b]) / 2.0)
    if not per:
        return torch.zeros(1, requires_grad=True, device=dev)
    return torch.stack(per).mean()


# ---------------------------------------------------------------------------------------------------------------------
# Visualisation names of src/utils.py:51-81 (out of scope as features): the trainer and testing.py import
# them by name (train_partseg_shapenet.py:6, testing.py:2, fitting.py:3,13), so they resolve here and do the file-less part of
# their job: no open3d, nothing is drawn.
# ---------------------------------------------------------------------------------------------------------------------
def save_point_cloud(filename, data):
    """src/utils.py:51-52: `np.savetxt(filename, data, delimiter=" ")`."""
    import numpy as np
    np.savetxt(filename, data.detach().cpu().numpy() if torch.is_tensor(data) else data, delimiter=" ")


def visualize_point_cloud(points, normals=[], colors=[], file="", viz=False):   # noqa: B006 (the reference's signature)
    """src/utils.py:55-72 builds an open3d point cloud, optionally draws and writes it.  Here: returns the arrays it was
    given as a dict (nothing is drawn; `viz=True` raises -- there is no display path in this package)."""
    if viz:
        raise NotImplementedError("visualize_point_cloud(viz=True): visualisation is out of scope (no open3d on this path)")
    return {"points": points, "normals": normals, "colors": colors}


def visualize_point_cloud_from_labels(points, labels, COLORS=None, normals=None, viz=False):
    """src/utils.py:75-81: colours the points by label and hands them to visualize_point_cloud."""
    import numpy as np
    lab = labels.detach().cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
    if COLORS is None:
        COLORS = np.random.rand(500, 3)
    return visualize_point_cloud(points, colors=COLORS[lab.astype(np.int64)], normals=normals if normals is not None else [], viz=viz)
