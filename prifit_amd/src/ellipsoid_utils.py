"""Call surface of the reference's src/ellipsoid_utils.py on the MI355X backend: guard_mean_shift (:9-27),
clustering (:31-73), sample_from_pred_params (:76-130), to_one_hot (:146-154), compute_approximate_ellipsoid_area (:157-159),
sample_from_pred_params_cuboid (:162-214)."""
import torch

from .. import fit_ops
from .._lib import call, cur_stream, ptr
from .sample_ellipsoid import SampleEllipsoid

MAXCLUSTERS = 25
sampleellipse = SampleEllipsoid()


def guard_mean_shift(embedding, number_samples, quantile, iterations, max_num_clusters, kernel_type="gaussian",
                     bandwidth_rows=None):
    """upstream :9-27 for one shape: embedding [N,D] -> (center [K,D], bandwidth, labels [N]); the quantile is doubled
    until at most `max_num_clusters` distinct labels remain."""
    if kernel_type != "gaussian":
        # the epanechnikov kernel (src/mean_shift.py:70-74): upstream's own loop over MeanShift.mean_shift -- per shape, not
        # on the loss path (the loss never passes kernel_type); the bandwidth subset, when given, is pinned for every retry
        from .mean_shift import MeanShift
        ms = MeanShift()
        while True:
            bw = None
            if bandwidth_rows is not None:
                with torch.no_grad():
                    bw = ms.compute_bandwidth(embedding, number_samples, quantile, rows=bandwidth_rows)
            center, bandwidth, cluster_ids = ms.mean_shift(embedding, number_samples, quantile, iterations,
                                                           kernel_type=kernel_type, bw=bw)
            if torch.unique(cluster_ids).shape[0] > max_num_clusters:
                quantile *= 2
            else:
                return center, bandwidth, cluster_ids
    rows = None if bandwidth_rows is None else torch.as_tensor(bandwidth_rows).reshape(1, -1)
    cl = fit_ops.cluster(embedding.unsqueeze(0).contiguous(), quantile, iterations, max_num_clusters,
                         num_samples=number_samples, bandwidth_rows=rows)
    K = int(cl["count"][0])
    return cl["centres"][0, :K], cl["bw"][0], cl["labels"][0]


def clustering(X, num_samples=1000, quantile=0.01, iterations=5, visualize=False, max_num_clusters=MAXCLUSTERS,
               bandwidth_rows=None, center_ids=None):
    """upstream :31-73.  X [B,N,D] unit rows -> (list of W_b [N,K_b], list of labels [N]).
    `num_samples` < N: the bandwidth is estimated on a row subset (upstream src/mean_shift.py:148-151) -- random per
    shape, or `bandwidth_rows` [B, num_samples] when the caller wants it reproducible."""
    cl = fit_ops.cluster(X.contiguous(), quantile, iterations, max_num_clusters, center_ids=center_ids,
                         num_samples=num_samples, bandwidth_rows=bandwidth_rows)
    counts = cl["count"].cpu().tolist()
    return [cl["W"][b, :, :counts[b]] for b in range(X.shape[0])], list(cl["labels"].unbind(0))


def to_one_hot(target, maxx=50):
    """upstream :146-154 (unused there): numpy integer labels [N] -> one-hot float [N, maxx] on the device."""
    idx = torch.as_tensor(target).to(device="cuda", dtype=torch.int64).unsqueeze(1)
    return torch.zeros(idx.shape[0], maxx, device="cuda").scatter_(1, idx, 1)


def compute_approximate_ellipsoid_area(a, b, c, p=1.585):
    """upstream :157-159 (3.142 and p = 1.585 kept verbatim)."""
    return 4 * 3.142 * ((a * b) ** p + (b * c) ** p + (c * a) ** p) ** (1 / p)


def sample_from_pred_params(ellipse_params_batch, N=500, batch_id=0, seed=0, visualize=False, class_list=[],
                            quantile=0.05):
    """upstream :76-130: per shape ~10000 surface points, shared among its ellipsoids in proportion to their approximate
    area (`round(10000 w_i)`, `<= 0 -> 100`; `N` is ignored upstream too), each ellipsoid sampled by
    SampleEllipsoid.sample.  Returns list[B] of [n_b, 3] tensors (gradients to r, V, centre), -1 for a shape without
    ellipsoids (:116)."""
    from .utils import pack_params
    first = next((p for prm in ellipse_params_batch for p in prm), None)
    if first is None:
        return [-1] * len(ellipse_params_batch)
    dev = first[0].device
    r, V, c, valid = pack_params(ellipse_params_batch, dev)
    B, K = valid.shape
    n = torch.empty(B, K, dtype=torch.int32, device=dev)
    off = torch.empty(B, K + 1, dtype=torch.int32, device=dev)
    call("prifit_sample_budget", ptr(r.detach().contiguous()), ptr(valid), B, K, fit_ops.sample_cap(K), ptr(n), ptr(off),
         cur_stream())
    n = n.cpu()
    out = []
    for b, prm in enumerate(ellipse_params_batch):
        pts = [sampleellipse.sample(p[0][0], p[0][1], p[0][2], p[2], p[1], n=int(n[b, k]))[0] for k, p in enumerate(prm)]
        out.append(torch.cat(pts, 0) if pts else -1)
    return out


def sample_from_pred_params_cuboid(ellipse_params_batch, N=500, batch_id=0, seed=0, visualize=False, class_list=[]):
    """upstream :162-214: the same for boxes with half-sides r -- budget proportional to the face area
    8 (ab + bc + ca) (:186-193), every box sampled by SampleEllipsoid.sample_cuboid."""
    from .utils import pack_params
    first = next((p for prm in ellipse_params_batch for p in prm), None)
    if first is None:
        return [-1] * len(ellipse_params_batch)
    dev = first[0].device
    r, V, c, valid = pack_params(ellipse_params_batch, dev)
    B, K = valid.shape
    n = torch.empty(B, K, dtype=torch.int32, device=dev)
    off = torch.empty(B, K + 1, dtype=torch.int32, device=dev)
    call("prifit_cuboid_sample_budget", ptr(r.detach().contiguous()), ptr(valid), B, K, fit_ops.sample_cap(K), ptr(n), ptr(off),
         cur_stream())
    n = n.cpu()
    out = []
    for b, prm in enumerate(ellipse_params_batch):
        pts = [sampleellipse.sample_cuboid(p[0][0], p[0][1], p[0][2], p[2], p[1], n=int(n[b, k]))[0] for k, p in enumerate(prm)]
        out.append(torch.cat(pts, 0) if pts else -1)
    return out
