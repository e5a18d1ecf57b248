"""Call surface of the reference's src/ellipsoid_utils.py on the MI355X backend."""
import torch

from .. import fit_ops

MAXCLUSTERS = 25


def clustering(X, num_samples=1000, quantile=0.01, iterations=5, visualize=False, max_num_clusters=MAXCLUSTERS):
    """upstream :31-73.  X [B,N,D] unit rows -> (list of W_b [N,K_b], list of labels [N])."""
    if num_samples != X.shape[1]:
        raise NotImplementedError("sub-sampled bandwidth estimation")
    cl = fit_ops.cluster(X.contiguous(), quantile, iterations, max_num_clusters)
    counts = cl["count"].cpu().tolist()
    return [cl["W"][b, :, :counts[b]] for b in range(X.shape[0])], list(cl["labels"].unbind(0))
