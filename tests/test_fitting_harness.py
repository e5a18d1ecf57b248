"""the stand-alone fitting harness of the reference, fitting.py:26-61 -- synthetic shapes made of 3
analytic ellipsoids x 500 points with 32-d one-hot embeddings -> clustering(X) with its DEFAULT arguments
(num_samples=1000 < N: sub-sampled bandwidth, quantile 0.01, 5 iterations) -> weighted_ellipsoid_fitting_batch ->
sample_from_pred_params -> Loss().loss -> backward.  Pinned by known answers (the generator needs trimesh upstream):
K = 3, recovered semi-axes within 2 %, centres, a small chamfer loss and a finite embedding gradient.
CPU: the oracle's restatement of the same flow.  GPU: the HIP backend through the reference's own module names
(prifit_amd.compat)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import prifit_oracle as orc
from prifit_amd import synth

B = 2


def _scene():
    pts, X, abc, ctr = synth.ellipsoid_scene(B, 91)
    rows = np.stack([np.random.default_rng(92 + b).permutation(pts.shape[1])[:1000] for b in range(B)])
    return torch.from_numpy(pts), torch.from_numpy(X), abc, ctr, torch.from_numpy(rows)


def _check(params, labels, abc, ctr, grad, loss):
    for b in range(B):
        assert len(params[b]) == 3, "K = %d" % len(params[b])
        lab = labels[b].cpu()
        assert sorted(torch.bincount(lab).tolist()) == [500, 500, 500]
        for k in range(3):
            # the cluster of generating ellipsoid k = the label its points carry
            kk = int(lab[k * 500])
            r, V, c = (t.detach().cpu().double().numpy() for t in params[b][kk])
            assert np.allclose(np.sort(r), np.sort(abc[b, k]), rtol=2e-2), (r, abc[b, k])
            assert np.abs(c - ctr[b, k]).max() < 0.02 * abc[b, k].max(), (c, ctr[b, k])
    assert torch.isfinite(grad).all()
    assert torch.isfinite(loss) and float(loss) < 0.5      # squared units of axes 2..19: the surfaces coincide


def test_fitting_harness_oracle():
    pts, X0, abc, ctr, rows = _scene()
    X_ = X0.clone().requires_grad_(True)
    X = F.normalize(X_, dim=2, p=2)
    Ws, labels, _ = orc.clustering(X, 0.01, 5, 25, bandwidth_rows=rows)
    R = torch.from_numpy(synth.uniform01((3, 3), 5))
    params = orc.fit_ellipsoids_batch(pts, Ws, [[R] * w.shape[1] for w in Ws])
    samples = orc.sample_from_params(params)
    per = []
    for b in range(B):
        s, t = samples[b], pts[b]
        d1 = ((t - s[orc.nearest_target(t, s)]) ** 2).sum(1).mean()
        d2 = ((s - t[orc.nearest_target(s, t)]) ** 2).sum(1).mean()
        per.append((d1 + d2) / 2)
    loss = torch.stack(per).mean()
    loss.backward()
    _check(params, labels, abc, ctr, X_.grad, loss)


@pytest.mark.gpu
def test_fitting_harness_hip_through_reference_module_names(hiplib):
    import importlib
    import sys
    from prifit_amd import compat
    saved = {k: sys.modules.get(k) for k in ("models", "src", "data_utils", "convex_loss", "testing")}
    try:
        compat.install()
        # the imports of fitting.py:1-18 that are not visualisation
        from src.fitting_utils import customsvd                      # noqa: F401
        from src.mean_shift import MeanShift                         # noqa: F401
        from src.guard import guard_exp                              # noqa: F401
        from src.sample_ellipsoid import SampleEllipsoid, Loss       # noqa: F401
        from src.ellipsoid_fitting import weighted_ellipsoid_fitting_batch
        from src.ellipsoid_utils import sample_from_pred_params, clustering
        from src.utils import analytic_chamfer_distance
        pts, X0, abc, ctr, rows = _scene()
        points = pts.cuda()
        X_ = X0.cuda().requires_grad_(True)
        X = F.normalize(X_, dim=2, p=2)
        weights_batch, labels = clustering(X)                        # defaults: num_samples=1000, quantile=0.01, 5 iterations
        params = weighted_ellipsoid_fitting_batch(points, weights_batch)
        resampled = sample_from_pred_params(params, 500)
        assert all(abs(r.shape[0] - 10000) <= 3 for r in resampled)
        loss = Loss().loss(points, resampled)
        loss.backward()
        _check(params, labels, abc, ctr, X_.grad.cpu(), loss.detach().cpu())
        # the list-based analytic chamfer distance (src/utils.py:384) agrees with the fused form of the training step
        from prifit_amd.convex_loss import analytic_chamfer_distance as fused
        from prifit_amd.src.utils import pack_params
        a = analytic_chamfer_distance(params, resampled, points)
        r, V, c, valid = pack_params(params, points.device)
        f, _ = fused(r, V, c, valid, points)
        assert abs(float(a) - float(f)) <= 1e-4 * abs(float(f)) + 1e-7
        # sub-sampled bandwidth with explicit rows = the oracle's on the same rows
        ms = MeanShift()
        for b in range(B):
            bw = ms.compute_bandwidth(X[b].detach(), 1000, 0.05, rows=rows[b])
            ref = orc.compute_bandwidth(F.normalize(X0[b], dim=1), 0.05, rows=rows[b])
            assert abs(float(bw) - float(ref)) <= 1e-5 * float(ref) + 1e-7
    finally:
        for k in [m for m in sys.modules if m.split(".")[0] in ("models", "src", "data_utils", "convex_loss", "testing")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
        importlib.invalidate_caches()
