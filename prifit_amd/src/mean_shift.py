"""Call surface of the reference's src/mean_shift.py (class MeanShift) on the MI355X backend.
Per-shape methods ([N, D] tensors, as upstream) are thin views over the batched kernels."""
import torch

from .. import fit_ops


class MeanShift:
    def mean_shift(self, X, num_samples, quantile, iterations, kernel_type="gaussian", bw=None, eff=False):
        """upstream :18-48 -> (center [K,D], bandwidth, labels [N])."""
        if kernel_type != "gaussian" or eff:
            raise NotImplementedError("only the gaussian, eff=False path is used by the reference's loss")
        Xb = X.unsqueeze(0).contiguous()
        if bw is None:
            with torch.no_grad():
                bw = self.compute_bandwidth(X, num_samples, quantile)
        bwb = torch.as_tensor(bw, dtype=torch.float32, device=X.device).reshape(1)
        N, D = X.shape
        rows = fit_ops.ROWS_BWD and fit_ops.rows_supported(N, D, fit_ops.KM)
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
        if rows and K <= fit_ops.KM:
            return fit_ops.MeanShiftRowsFn.apply(Xb, bwb, ids, None, traj)[0], bw, labels[0].long()
        if rows:   # more kept centres than row slots: the dense engine
            Zg = fit_ops.MeanShiftFn.apply(Xb, bwb, iterations)
        return Zg[0][ids[0]], bw, labels[0].long()

    def mean_shift_(self, X, b, iterations=10, kernel_type="gaussian"):
        """upstream :50-84 -> (new_X, X)."""
        bwb = torch.as_tensor(b, dtype=torch.float32, device=X.device).reshape(1)
        return fit_ops.MeanShiftFn.apply(X.unsqueeze(0).contiguous(), bwb, iterations)[0], X

    def compute_bandwidth(self, X, num_samples, quantile, rows=None):
        """upstream :138-160; num_samples < N takes the statistic over a random row subset (`rows` [num_samples]
        makes it reproducible)."""
        rows = None if rows is None else torch.as_tensor(rows).reshape(1, -1)
        return fit_ops.compute_bandwidth(X.unsqueeze(0).contiguous(), quantile, num_samples, rows)[0]

    def nms(self, centers, X, b):
        """upstream :162-202 for centers is X (the only way it is called, :44)."""
        if centers is not X and not (centers.shape == X.shape and centers.data_ptr() == X.data_ptr()):
            raise NotImplementedError("nms(centers, X, b) is implemented for centers is X only (upstream calls it as "
                                      "nms(new_X, new_X, b), src/mean_shift.py:44)")
        bwb = torch.as_tensor(b, dtype=torch.float32, device=X.device).reshape(1)
        ids, count, labels, _ = fit_ops.nms(X.unsqueeze(0).contiguous(), bwb)
        K = int(count.item())
        ids = ids[0, :K].long()
        return centers[ids], ids, labels[0].long()

    def membership(self, centers, X, bandwidth):
        """upstream :230-247 -> [K, N]."""
        K = centers.shape[0]
        cpad = torch.zeros(1, fit_ops.KM, X.shape[1], device=X.device)
        cpad[0, :K] = centers
        W = fit_ops.MembershipFn.apply(cpad, X.unsqueeze(0).contiguous(),
                                       torch.as_tensor(bandwidth, dtype=torch.float32, device=X.device).reshape(1),
                                       torch.tensor([K], dtype=torch.int32, device=X.device))
        return W[0, :, :K].t()
