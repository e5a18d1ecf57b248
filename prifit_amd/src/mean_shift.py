"""Call surface of the reference's src/mean_shift.py (class MeanShift) on the MI355X backend.
Per-shape methods ([N, D] tensors, as upstream) are thin views over the batched kernels.

The gaussian kernel with eff=False is the path the loss takes (src/ellipsoid_utils.py:19-22) and the one the HIP kernels
implement.  The two variants upstream never calls -- the epanechnikov kernel (:70-74) and `mean_shift_eff_` (:86-136,
eff=True) -- are provided for completeness of the call surface as plain device-tensor arithmetic (three matrix products per
iteration through autograd; not tuned, not on the hot path), pinned against the reference by
tests/golden/fit_meanshift_variants.npz."""
import numpy as np
import torch

from .. import fit_ops
from .._lib import require_cuda


def _guard_exp(x):
    return torch.exp(torch.clamp(x, min=-13.0, max=75.0))      # src/guard.py:6-11


def _normalize_rows(Z):
    return Z / torch.norm(Z, dim=1, p=2, keepdim=True)


class MeanShift:
    def mean_shift(self, X, num_samples, quantile, iterations, kernel_type="gaussian", bw=None, eff=False, seed_rows=None):
        """upstream :18-48 -> (center [K,D], bandwidth, labels).  eff=True (:33-39): half of the points, chosen at random
        upstream (`np.random.choice`) or given as `seed_rows`, are shifted over the full dictionary; the labels are then
        those of the seed points."""
        Xb = X.unsqueeze(0).contiguous()
        if bw is None:
            with torch.no_grad():
                bw = self.compute_bandwidth(X, num_samples, quantile)
        bwb = torch.as_tensor(bw, dtype=torch.float32, device=X.device).reshape(1)
        if eff or kernel_type != "gaussian":
            if eff:
                rows = (np.random.choice(X.shape[0], X.shape[0] // 2, replace=False) if seed_rows is None else seed_rows)
                rows = torch.as_tensor(rows, dtype=torch.long, device=X.device)
                new_X, _ = self.mean_shift_eff_(X, X[rows], b=bw, iterations=iterations, kernel_type=kernel_type)
            else:
                new_X, _ = self.mean_shift_(X, b=bw, iterations=iterations, kernel_type=kernel_type)
            with torch.no_grad():
                _, indices, labels = self.nms(new_X.detach(), new_X.detach(), b=bw)
            return new_X[indices], bw, labels
        N, D = X.shape
        rows = fit_ops.ROWS_BWD and fit_ops.rows_supported(N, D, fit_ops.KM_MAX)
        if rows:   # the gradient enters through `new_X[indices]` (:46) alone: trajectory now, row-sparse backward later
            with torch.no_grad():
                Z, traj = fit_ops.mean_shift_trajectory(Xb.detach(), bwb, iterations, keep_kernel=False)
        else:
            Zg = fit_ops.MeanShiftFn.apply(Xb, bwb, iterations)
            Z = Zg.detach()
        with torch.no_grad():
            ids, count, labels, _ = fit_ops.nms(Z, bwb)
        K = int(count.item())
        if K > fit_ops.NMS_CAP:
            raise RuntimeError("more than %d clusters" % fit_ops.NMS_CAP)
        ids = ids[:, :K].long()
        if rows and K <= fit_ops.KM_MAX:
            return fit_ops.MeanShiftRowsFn.apply(Xb, bwb, ids, None, traj)[0], bw, labels[0].long()
        if rows:   # more kept centres than row slots: the dense engine
            Zg = fit_ops.MeanShiftFn.apply(Xb, bwb, iterations)
        return Zg[0][ids[0]], bw, labels[0].long()

    def mean_shift_(self, X, b, iterations=10, kernel_type="gaussian"):
        """upstream :50-84 -> (new_X, X)."""
        require_cuda(X)
        if kernel_type != "gaussian":        # epanechnikov (:70-74)
            b = torch.as_tensor(b, dtype=X.dtype, device=X.device)
            new_X = X.clone()
            for _ in range(iterations):
                dist = 2.0 - 2.0 * new_X @ X.t()
                K = torch.relu(3 / 4 * (1 - dist / (b ** 2)))
                D = 1 / torch.sum(K, 1, keepdim=True)
                new_X = _normalize_rows(new_X + ((K @ X) * D - new_X))
            return new_X, X
        bwb = torch.as_tensor(b, dtype=torch.float32, device=X.device).reshape(1)
        return fit_ops.MeanShiftFn.apply(X.unsqueeze(0).contiguous(), bwb, iterations)[0], X

    def mean_shift_eff_(self, X, X_seed, b, iterations=10, kernel_type="gaussian"):
        """upstream :86-136 -> (X_seed shifted, X).  (The gaussian branch's exponent is `X_seed X^T / b^2` upstream.)"""
        require_cuda(X, X_seed)
        b = torch.as_tensor(b, dtype=X.dtype, device=X.device)
        for _ in range(iterations):
            if kernel_type == "gaussian":
                K = _guard_exp((X_seed @ X.t()) / (b ** 2))
            else:
                K = torch.relu(3 / 4 * (1 - (2.0 - 2.0 * X_seed @ X.t()) / (b ** 2)))
            D = 1 / torch.sum(K, 1, keepdim=True)
            X_seed = _normalize_rows((K @ X) * D)
        return X_seed, X

    def compute_bandwidth(self, X, num_samples, quantile, rows=None):
        """upstream :138-160; num_samples < N takes the statistic over a random row subset (`rows` [num_samples]
        makes it reproducible)."""
        rows = None if rows is None else torch.as_tensor(rows).reshape(1, -1)
        return fit_ops.compute_bandwidth(X.unsqueeze(0).contiguous(), quantile, num_samples, rows)[0]

    def nms(self, centers, X, b):
        """upstream :162-202 -> (centers[ids], ids ascending, labels [N]).  `centers is X` is how upstream calls it (:44,
        :39); two different tables (the commented call `nms(new_X, X, b)` of :43) must have the same row count, as upstream's
        broadcast at :191 requires."""
        require_cuda(centers, X)
        bwb = torch.as_tensor(b, dtype=torch.float32, device=X.device).reshape(1)
        if centers is X or (centers.shape == X.shape and centers.data_ptr() == X.data_ptr()):
            ids, count, labels, _ = fit_ops.nms(X.unsqueeze(0).contiguous(), bwb)
        else:
            if centers.shape != X.shape:
                raise RuntimeError("nms(centers, X, b): centers %s and X %s must have the same shape (upstream's "
                                   "cluster_nbrs[uniques] * num_mem_cluster broadcast, src/mean_shift.py:191)"
                                   % (tuple(centers.shape), tuple(X.shape)))
            ids, count, labels, _ = fit_ops.nms_pair(centers.detach().unsqueeze(0).contiguous(),
                                                     X.detach().unsqueeze(0).contiguous(), bwb)
        K = int(count.item())
        if K > fit_ops.NMS_CAP:
            raise RuntimeError("more than %d clusters" % fit_ops.NMS_CAP)
        ids = ids[0, :K].long()
        return centers[ids], ids, labels[0].long()

    def membership(self, centers, X, bandwidth):
        """upstream :230-247 -> [K, N]."""
        K = centers.shape[0]
        cpad = torch.zeros(1, fit_ops.slots_for(K), X.shape[1], device=X.device)
        cpad[0, :K] = centers
        W = fit_ops.MembershipFn.apply(cpad, X.unsqueeze(0).contiguous(),
                                       torch.as_tensor(bandwidth, dtype=torch.float32, device=X.device).reshape(1),
                                       torch.tensor([K], dtype=torch.int32, device=X.device))
        return W[0, :, :K].t()
