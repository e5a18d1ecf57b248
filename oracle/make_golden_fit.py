"""Fitting half of oracle/make_golden.py: mean-shift / nms / membership / ellipsoid fit / SDF /
analytic chamfer / convex_loss -- oracle vs the imported reference, and golden fixtures.

TEST INFRASTRUCTURE (build container only).  Harness conventions shared by reference and oracle
so that results are comparable (SURVEY.md 8a' q14, q17, q19, 8c):
  * the covariance noise `torch.rand(3,3)` (src/ellipsoid_fitting.py:38) is replaced by ONE fixed
    matrix R for every cluster (cluster order is rounding noise, so per-cluster noise cannot be shared);
  * SVD column signs are pinned by `canonical_signs` on both sides;
  * the trimesh surface sampler is replaced by the build's Fibonacci (U,V) table on both sides;
  * clusters are compared in a partition-canonical order (ascending smallest member index).
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import refshim  # noqa: E402
import prifit_oracle as orc  # noqa: E402
from prifit_amd import synth  # noqa: E402


def fit_inputs(B, N, D, seed, M=5000, noise=0.03):
    """Blob cloud (targets M points, model input = fixed N-subset) + prototype embedding.
    noise=0.03 gives 8 clusters with SOFT memberships (max weight ~0.9), i.e. non-trivial gradients;
    the survey's 0.01 saturates the membership clamp and leaves gradients of 1e-8."""
    cham, lab = synth.blobs_with_labels(B, M, seed)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    pts = cham[:, sel]
    emb = synth.prototype_embedding(lab[:, sel], D, seed + 2, noise=noise)
    return torch.from_numpy(pts), torch.from_numpy(cham), torch.from_numpy(emb)


def canonical_order(labels, K):
    """Order of clusters by ascending smallest member index (partition-canonical)."""
    first = []
    for k in range(K):
        idx = torch.nonzero(labels == k)
        first.append(idx.min().item() if idx.numel() else 10 ** 9)
    return torch.tensor(np.argsort(np.array(first), kind="stable"))


def same_partition(la, lb):
    """True if two labelings induce the same partition."""
    pairs = torch.unique(torch.stack([la, lb], 1), dim=0)
    return pairs.shape[0] == torch.unique(la).shape[0] == torch.unique(lb).shape[0]


@contextlib.contextmanager
def patched(obj, name, value):
    old = getattr(obj, name)
    setattr(obj, name, value)
    try:
        yield
    finally:
        setattr(obj, name, old)


def run_bandwidth_over(save, close):
    """num_samples > N (src/mean_shift.py:151-155: `X[L[0:num_samples]]` keeps all N rows, K = int(quantile *
    num_samples)): clustering(X)'s default num_samples = 1000 on a 512-point cloud.  Own fixture, own entry
    (`make_golden.py bw_over`), so that the other fitting fixtures are not rewritten."""
    ms = refshim.ref("src.mean_shift").MeanShift()
    seed, B, N, D, q = 2, 2, 512, 128, 0.05
    _, _, emb = fit_inputs(B, N, D, seed)
    out = {"seed": seed, "N": N, "num_samples": 1000, "quantile": q}
    for b in range(B):
        with torch.no_grad():
            bw_r = ms.compute_bandwidth(emb[b], 1000, q)
        bw_o = orc.compute_bandwidth(emb[b], q, num_samples=1000)
        close(bw_o, bw_r, f"bandwidth with num_samples > N b={b}", rtol=1e-6)
        assert abs(float(orc.compute_bandwidth(emb[b], q)) - float(bw_r)) > 1e-3 * float(bw_r)   # k = 25 is another statistic
        out[f"bw_{b}"] = bw_r
    save("fit_bandwidth_over", **out)


def run_mean_shift_variants(save, close):
    """The two variants of MeanShift the loss never takes (src/mean_shift.py:70-74 epanechnikov kernel, :86-136
    mean_shift_eff_): oracle vs reference, forward and d/dX.  Own fixture (`make_golden.py ms_variants`)."""
    ms = refshim.ref("src.mean_shift").MeanShift()
    seed, N, D = 6, 512, 32
    _, _, emb = fit_inputs(1, N, D, seed, M=1000, noise=0.1)
    X0 = emb[0]
    G = torch.from_numpy(synth.features(1, N, D, seed + 1))[0]
    seeds = torch.from_numpy(np.random.default_rng(seed).choice(N, N // 2, replace=False))
    out = {"seed": seed, "rows": seeds.numpy().astype(np.int16)}
    b = torch.tensor(0.9)
    for name, fr, fo, g in (
            ("epa", lambda X: ms.mean_shift_(X, b, 3, kernel_type="epa")[0], lambda X: orc.mean_shift_iterations(X, b, 3, "epa"), G),
            ("eff", lambda X: ms.mean_shift_eff_(X, X[seeds], b, 3)[0], lambda X: orc.mean_shift_eff(X, X[seeds], b, 3), G[: N // 2]),
            ("eff_epa", lambda X: ms.mean_shift_eff_(X, X[seeds], b, 3, kernel_type="epa")[0],
             lambda X: orc.mean_shift_eff(X, X[seeds], b, 3, "epa"), G[: N // 2])):
        Xr = X0.clone().requires_grad_(True)
        Zr = fr(Xr)
        (Zr * g).sum().backward()
        Xo = X0.clone().requires_grad_(True)
        Zo = fo(Xo)
        (Zo * g).sum().backward()
        close(Zo.detach(), Zr.detach(), "mean-shift variant %s: Z" % name, rtol=1e-5, atol=1e-6)
        close(Xo.grad, Xr.grad, "mean-shift variant %s: dX" % name, rtol=1e-4, atol=1e-6 * Xr.grad.abs().max().item())
        out["Z_" + name] = Zr.detach()[:64]
        out["Zsum_" + name] = Zr.detach().sum(0)
        out["dX_" + name] = Xr.grad[:64]
        out["dXnorm_" + name] = Xr.grad.norm()
    save("fit_meanshift_variants", **out)


def run_nms_pair_and_epa_guard(save, eq, close):
    """Two corners of the reference's call surface the loss never takes (own entry `make_golden.py nms_pair`):
    (i) MeanShift.nms(centers, X, b) with centres that are NOT the points (src/mean_shift.py:162-202; the commented call
    `nms(new_X, X, b)` of :43): shifted points against the original embedding; (ii) guard_mean_shift with the
    epanechnikov kernel (src/ellipsoid_utils.py:9-27 forwards `kernel_type`, src/mean_shift.py:70-74).  Oracle vs
    reference exactly (the same CPU arithmetic); the fixture stores K, ids and labels."""
    ms = refshim.ref("src.mean_shift").MeanShift()
    EU = refshim.ref("src.ellipsoid_utils")
    seed, N, D = 9, 512, 32
    _, _, emb = fit_inputs(2, N, D, seed, M=1000, noise=0.1)
    out = {"seed": seed}
    for b in range(2):
        X = emb[b]
        with torch.no_grad():
            bw = ms.compute_bandwidth(X, N, 0.05)
            Z = ms.mean_shift_(X, bw, 4)[0]              # part of the way: the modes have not collapsed to a point yet
            cr, ir, lr = ms.nms(Z, X, bw)
        co, io, lo = orc.nms(Z, X, bw)
        eq(io, ir, "nms(Z, X) ids b=%d (K = %d)" % (b, ir.shape[0]))
        eq(lo, lr, "nms(Z, X) labels b=%d" % b)
        assert 2 <= ir.shape[0] <= 32, ir.shape
        out["bw_%d" % b] = bw
        out["ids_%d" % b] = ir.to(torch.int16)
        out["labels_%d" % b] = lr.to(torch.int16)
    # (ii) epanechnikov through guard_mean_shift, q small enough that the first pass finds more than the cap
    X = emb[0]
    q0, cap, iters = 0.02, 6, 5
    calls = []
    real = EU.meanshift.mean_shift

    def counting(*a, **k):
        calls.append(a[2])
        return real(*a, **k)

    with patched(EU.meanshift, "mean_shift", counting):
        cr, bwr, lr = EU.guard_mean_shift(X, N, q0, iters, cap, kernel_type="epa")
    co, bwo, lo, io, Zo, qo = orc.guard_mean_shift(X, q0, iters, cap, kernel_type="epa")
    close(bwo, bwr, "guard_mean_shift(epa) bandwidth", rtol=1e-6)
    assert qo == calls[-1], (qo, calls)
    eq(lo, lr, "guard_mean_shift(epa) labels")
    close(co, cr, "guard_mean_shift(epa) centres", rtol=1e-5, atol=1e-6)
    print("  quantiles tried by the reference:", calls, " K =", cr.shape[0])
    assert len(calls) >= 2, "pick q0 / cap so that a retry happens"
    out.update(q0=q0, cap=cap, iters=iters, epa_quantiles=np.array(calls), epa_bw=bwr, epa_K=cr.shape[0],
               epa_labels=lr.to(torch.int16))
    save("fit_nms_pair", **out)


def run_center_grad(save, eq, close):
    """The gradient that enters the mean-shift trajectory through `center = new_X[indices]` ALONE (src/mean_shift.py:44-46:
    `nms` runs under no_grad, the gather is the only differentiable use of the shifted points): d/dX sum(G * new_X[ids])
    from the reference's autograd through all ten updates.  Pins the row-sparse backward engine (prifit_amd
    fit_ops.MeanShiftRowsFn) directly.  Own fixture (`make_golden.py center_grad`)."""
    ms = refshim.ref("src.mean_shift").MeanShift()
    B, N, D, seed, q, iters = 2, 2048, 128, 31, 0.05, 10
    _, _, emb = fit_inputs(B, N, D, seed)
    out = {"seed": seed, "quantile": q, "iterations": iters}
    for b in range(B):
        X = emb[b]
        with torch.no_grad():
            bw = ms.compute_bandwidth(X, N, q)
        Xr = X.clone().requires_grad_(True)
        Zr, _ = ms.mean_shift_(Xr, bw, iterations=iters)
        with torch.no_grad():
            _, ids, _ = ms.nms(Zr.detach(), Zr.detach(), bw)
        Gc = torch.from_numpy(synth.features(1, ids.shape[0], D, seed + 11 + b))[0]
        (Zr[ids] * Gc).sum().backward()                     # center = new_X[indices]; only that gather reaches the loss
        Xo = X.clone().requires_grad_(True)
        Zo = orc.mean_shift_iterations(Xo, bw, iters)
        (Zo[ids] * Gc).sum().backward()
        close(Zo[ids].detach(), Zr[ids].detach(), f"centres b={b}", rtol=1e-5, atol=1e-6)
        close(Xo.grad, Xr.grad, f"d/dX sum(G * new_X[ids]) b={b}", rtol=1e-3, atol=1e-5 * Xr.grad.abs().max().item())
        g = Xr.grad
        out[f"bw_{b}"] = bw
        out[f"ids_{b}"] = ids.to(torch.int16)
        out[f"G_{b}"] = Gc
        out[f"centres_{b}"] = Zr[ids].detach()
        out[f"dX_head_{b}"] = g[:64]
        out[f"dX_ids_{b}"] = g[ids]
        out[f"dX_colsum_{b}"] = g.sum(0)
        out[f"dX_rownorm_{b}"] = g.norm(dim=1)
        out[f"dX_norm_{b}"] = g.norm()
    save("fit_center_grad", **out)


def run_prune(save, eq, close):
    """prune_points (convex_loss.py:444-470) on overlapping ellipsoids: oracle vs reference, and the fixture the HIP path is
    tested against (`make_golden.py prune`).  Surface points come from the shared Fibonacci table (oracle sampler)."""
    CL = refshim.ref("convex_loss")
    rng = np.random.default_rng(77)
    params_batch = []
    for b in range(3):
        K = (3, 5, 1)[b]
        params = []
        for k in range(K):
            r = torch.from_numpy(rng.uniform(0.15, 0.5, 3).astype(np.float32))
            Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            if np.linalg.det(Q) < 0:
                Q[:, 2] = -Q[:, 2]
            c = torch.from_numpy(rng.uniform(-0.25, 0.25, 3).astype(np.float32))       # close centres: heavy overlap
            params.append((r, torch.from_numpy(Q.astype(np.float32)), c))
        params_batch.append(params)
    points = orc.sample_from_params(params_batch)
    ref = CL.prune_points(points, params_batch)
    mine = orc.prune_points(points, params_batch)
    out = {}
    for b, (pr, po) in enumerate(zip(ref, mine)):
        eq(po, pr, f"prune_points b={b}")
        with torch.no_grad():
            sdf = torch.stack(CL.compute_sdf_ellipsoids(points[b], params_batch[b]), 1)
        m = sdf.min(1)[0]
        assert 0 < pr.shape[0] <= points[b].shape[0]
        out[f"r_{b}"] = torch.stack([p[0] for p in params_batch[b]])
        out[f"V_{b}"] = torch.stack([p[1] for p in params_batch[b]])
        out[f"c_{b}"] = torch.stack([p[2] for p in params_batch[b]])
        out[f"n_{b}"] = np.int32(points[b].shape[0])
        out[f"keep_{b}"] = (m > -1e-3).numpy()
        out[f"minsdf_{b}"] = m
        out[f"kept_sum_{b}"] = pr.sum(0)
    print("    kept %s of %s" % ([int(out[f"keep_{b}"].sum()) for b in range(3)], [int(out[f"n_{b}"]) for b in range(3)]))
    save("fit_prune", **out)


def overlapping_params(seed, Ks, grad=False):
    """Ellipsoids with close centres (heavy overlap), one list per shape."""
    rng = np.random.default_rng(seed)
    batch = []
    for K in Ks:
        params = []
        for k in range(K):
            r = torch.from_numpy(rng.uniform(0.15, 0.5, 3).astype(np.float32))
            Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            if np.linalg.det(Q) < 0:
                Q[:, 2] = -Q[:, 2]
            c = torch.from_numpy(rng.uniform(-0.25, 0.25, 3).astype(np.float32))
            params.append(tuple(t.requires_grad_(grad) for t in (r, torch.from_numpy(Q.astype(np.float32)), c)))
        batch.append(params)
    return batch


def run_intersections(save, eq, close):
    """The intersection-loss variants upstream keeps but does not call (convex_loss.py:106, :163, :227, :346, :416) and
    sample_axis (:285): oracle vs reference, value and gradient with respect to every (r, V, c); fixture for the HIP-side
    adapters (`make_golden.py intersections`)."""
    CL = refshim.ref("convex_loss")
    Ks = (3, 5, 1, 2)
    rng = np.random.default_rng(91)
    surf = [torch.from_numpy(rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)) for n in (300, 257, 64, 128)]
    pts = torch.from_numpy(rng.uniform(-0.6, 0.6, (len(Ks), 256, 3)).astype(np.float32))
    out = {"Ks": np.array(Ks, np.int32), "pts": pts}
    for b, p in enumerate(surf):
        out[f"surf_{b}"] = p
    base = overlapping_params(78, Ks)
    for b in range(len(Ks)):
        out[f"r_{b}"] = torch.stack([p[0] for p in base[b]])
        out[f"V_{b}"] = torch.stack([p[1] for p in base[b]])
        out[f"c_{b}"] = torch.stack([p[2] for p in base[b]])
    r0, V0, c0 = base[0][1]
    ax_r, ax_o = CL.sample_axis(r0, V0, c0), orc.sample_axis(r0, V0, c0)
    close(ax_o, ax_r, "sample_axis", rtol=1e-6, atol=1e-7)
    out["axis_samples"] = ax_r
    variants = {
        "surface": (lambda P: CL.compute_intersection_loss(P, surf), lambda P: orc.intersection_loss_surface(P, surf)),
        "surface_cuboid": (lambda P: CL.compute_intersection_loss_cuboid(P, surf),
                           lambda P: orc.intersection_loss_surface(P, surf, cuboid=True)),
        "volume": (lambda P: CL.compute_intersection_loss_volume(P, surf), lambda P: orc.intersection_loss_volume(P, surf)),
        "volume_2": (lambda P: CL.compute_intersection_loss_volume_2(P, pts), lambda P: orc.intersection_loss_volume_2(P, pts)),
        "volume_4": (lambda P: CL.compute_intersection_loss_volume_4(P, pts), lambda P: orc.intersection_loss_volume_4(P, pts)),
    }
    for name, (fr, fo) in variants.items():
        Pr, Po = overlapping_params(78, Ks, grad=True), overlapping_params(78, Ks, grad=True)
        lr, lo = fr(Pr), fo(Po)
        close(lo, lr, f"intersection {name}", rtol=1e-5, atol=1e-8)
        lr.backward()
        lo.backward()
        out[f"{name}_loss"] = lr.detach().reshape(())
        for b in range(len(Ks)):
            for i, tag in enumerate("rVc"):
                gr = torch.stack([torch.zeros_like(p[i]) if p[i].grad is None else p[i].grad for p in Pr[b]])
                go = torch.stack([torch.zeros_like(p[i]) if p[i].grad is None else p[i].grad for p in Po[b]])
                close(go, gr, f"intersection {name} d{tag} b={b}", rtol=1e-4, atol=1e-6 * max(1.0, gr.abs().max().item()))
                out[f"{name}_d{tag}_{b}"] = gr
        print("    %-15s loss %.6e" % (name, lr.item()))
    empty = CL.compute_intersection_loss([], [])
    assert empty.shape == (1,) and empty.item() == 0.0
    save("fit_intersections", **out)


def run(save, eq, close):
    print("[fit]")
    MS = refshim.ref("src.mean_shift")
    EF = refshim.ref("src.ellipsoid_fitting")
    EU = refshim.ref("src.ellipsoid_utils")
    CL = refshim.ref("convex_loss")
    UT = refshim.ref("src.utils")
    ms = MS.MeanShift()
    B, N, D, seed = 2, 2048, 128, 31
    pts, cham, emb = fit_inputs(B, N, D, seed)
    R = torch.from_numpy(synth.uniform01((3, 3), seed))
    q, iters = 0.05, 10

    # ---- bandwidth / mean-shift iterations / nms / membership, per shape
    out = {"seed": seed, "R": R}
    G = torch.from_numpy(synth.features(B, N, D, seed + 3))
    for b in range(B):
        X = emb[b]
        with torch.no_grad():
            bw_r = ms.compute_bandwidth(X, N, q)
        bw_o = orc.compute_bandwidth(X, q)
        close(bw_o, bw_r, f"bandwidth b={b}", rtol=1e-6)
        Xr = X.clone().requires_grad_(True)
        Zr, _ = ms.mean_shift_(Xr, bw_r, iterations=iters)
        (Zr * G[b]).sum().backward()
        Xo = X.clone().requires_grad_(True)
        Zo = orc.mean_shift_iterations(Xo, bw_r, iters)
        (Zo * G[b]).sum().backward()
        close(Zo, Zr, f"mean_shift_ Z b={b}", rtol=1e-5, atol=1e-6)
        close(Xo.grad, Xr.grad, f"mean_shift_ dX b={b}", rtol=1e-3, atol=1e-5 * Xr.grad.abs().max().item())
        with torch.no_grad():
            cr, ir, lr = ms.nms(Zr.detach(), Zr.detach(), bw_r)
            co, io, lo = orc.nms(Zr.detach(), Zr.detach(), bw_r)
        eq(io, ir, f"nms ids b={b} (same input)")
        eq(lo, lr, f"nms labels b={b} (same input)")
        Wr = ms.membership(cr, X, bw_r)
        Wo = orc.membership(co, X, bw_r)
        close(Wo, Wr, f"membership b={b}", rtol=1e-5, atol=1e-7)
        out[f"bw_{b}"] = bw_r
        out[f"Z_head_{b}"] = Zr.detach()[:64]
        out[f"Z_colsum_{b}"] = Zr.detach().sum(0)
        out[f"dX_head_{b}"] = Xr.grad[:64]
        out[f"dX_norm_{b}"] = Xr.grad.norm()
        out[f"K_{b}"] = ir.shape[0]
        out[f"labels_{b}"] = lr.to(torch.int16)
        order = canonical_order(lr, ir.shape[0])
        out[f"W_colsum_{b}"] = Wr.detach().t()[:, order].sum(0)
        out[f"W_head_{b}"] = Wr.detach().t()[:64][:, order]
    save("fit_meanshift", **out)

    # ---- sub-sampled bandwidth (src/mean_shift.py:148-151, num_samples < N: the default of clustering(X), fitting.py:43).
    # The reference draws the subset with np.random.shuffle; the harness seeds numpy, replays the shuffle and hands the
    # same rows to the oracle.
    sub = {"seed": seed}
    for b in range(B):
        np.random.seed(500 + b)
        L = np.arange(N)
        np.random.shuffle(L)
        rows = L[:1000].copy()
        np.random.seed(500 + b)
        with torch.no_grad():
            bw_r = ms.compute_bandwidth(emb[b], 1000, q)
        bw_o = orc.compute_bandwidth(emb[b], q, rows=rows)
        close(bw_o, bw_r, f"sub-sampled bandwidth b={b}", rtol=1e-6)
        sub[f"rows_{b}"] = rows.astype(np.int16)
        sub[f"bw_{b}"] = bw_r
    save("fit_bandwidth_sub", **sub)

    # ---- ellipsoid fit: soft weights from the clustering + a hard one-hot known-answer case
    def ref_customsvd_canonical(M):
        U, S, V = refshim.ref("src.fitting_utils").customsvd(M)
        s = orc.canonical_signs(V).view(1, 3)
        return U * s, S, V * s

    Ws_r, labs_r = EU.clustering(emb, num_samples=N, quantile=q, iterations=iters, max_num_clusters=25)
    Ws_o, labs_o, info_o = orc.clustering(emb, q, iters, 25)
    out = {"seed": seed, "R": R}
    for b in range(B):
        assert same_partition(labs_r[b], labs_o[b]), "label partition differs"
        print(f"  ok partition      clustering labels b={b} (K={Ws_r[b].shape[1]})")
    Wr = [w.detach().clone().requires_grad_(True) for w in Ws_r]
    with patched(torch, "rand", lambda *a, **k: R.clone()), patched(EF, "customsvd", ref_customsvd_canonical):
        params_r = EF.weighted_ellipsoid_fitting_batch(pts, Wr)
    Wo = [w.detach().clone().requires_grad_(True) for w in Ws_r]
    table = [[R] * w.shape[1] for w in Wo]
    params_o = orc.fit_ellipsoids_batch(pts, Wo, table, canonical=True)
    gr = torch.from_numpy(synth.features(1, 32, 15, seed + 4))[0]
    for b in range(B):
        assert len(params_r[b]) == len(params_o[b]) == Wr[b].shape[1]
        lr_ = sum((p[0] * gr[k, 0:3]).sum() + (p[1] * gr[k, 3:12].view(3, 3)).sum() + (p[2] * gr[k, 12:15]).sum()
                  for k, p in enumerate(params_r[b]))
        lo_ = sum((p[0] * gr[k, 0:3]).sum() + (p[1] * gr[k, 3:12].view(3, 3)).sum() + (p[2] * gr[k, 12:15]).sum()
                  for k, p in enumerate(params_o[b]))
        lr_.backward()
        lo_.backward()
        for k, (a, o) in enumerate(zip(params_r[b], params_o[b])):
            close(o[0], a[0], f"fit r b={b} k={k}", rtol=1e-5, atol=1e-6)
            close(o[1], a[1], f"fit V b={b} k={k}", rtol=1e-4, atol=1e-5)
            close(o[2], a[2], f"fit c b={b} k={k}", rtol=1e-5, atol=1e-6)
        close(Wo[b].grad, Wr[b].grad, f"fit dW b={b}", rtol=1e-3, atol=1e-5 * Wr[b].grad.abs().max().item())
        out[f"W_{b}"] = Ws_r[b].detach().to(torch.float32)
        out[f"r_{b}"] = torch.stack([p[0] for p in params_r[b]]).detach()
        out[f"V_{b}"] = torch.stack([p[1] for p in params_r[b]]).detach()
        out[f"c_{b}"] = torch.stack([p[2] for p in params_r[b]]).detach()
        out[f"dW_{b}"] = Wr[b].grad
    out["grad_seed_table"] = gr
    save("fit_ellipsoid", **out)

    # ---- SDF + analytic chamfer with the shared sampler
    def ref_sample(self, a, b_, c, center, transformation, n=500):
        U, V = orc.fibonacci_uv(int(n))
        p = self.uniform_sample_points_on_ellipsoid(U, V, a, b_, c)
        return p @ transformation.T + center, None

    SE = refshim.ref("src.sample_ellipsoid")
    with patched(SE.SampleEllipsoid, "sample", ref_sample):
        samples_r = EU.sample_from_pred_params(params_r, 500)
    samples_o = orc.sample_from_params(params_o)
    for b in range(B):
        close(samples_o[b], samples_r[b], f"sampled points b={b}", rtol=1e-5, atol=1e-6)
    l_r = UT.analytic_chamfer_distance(params_r, samples_r, cham)
    l_o, parts = orc.analytic_chamfer(params_o, samples_o, cham)
    close(l_o, l_r, "analytic chamfer", rtol=1e-5)
    sdf_r = CL.compute_sdf_ellipsoids_batch(cham, params_r)
    save("fit_chamfer", seed=seed, loss=l_r.detach(), dist_st=torch.stack([p[0] for p in parts]),
         sdf_ts=torch.stack([p[1] for p in parts]), nsamples=np.array([s.shape[0] for s in samples_r]),
         sdf_head=torch.stack([torch.stack(s, 1)[:256] for s in sdf_r]).detach())

    # ---- full convex_loss (B=2), gradient w.r.t. the embedding
    Xr = emb.permute(0, 2, 1).clone().requires_grad_(True)
    with patched(torch, "rand", lambda *a, **k: R.clone()), patched(EF, "customsvd", ref_customsvd_canonical), \
            patched(SE.SampleEllipsoid, "sample", ref_sample):
        tot_r, ch_r, prm_r, lab_r = CL.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xr, quantile=q,
                                                   iterations=iters, max_num_clusters=25)
    tot_r.sum().backward()
    Xo = emb.permute(0, 2, 1).clone().requires_grad_(True)
    tot_o, ch_o, prm_o, lab_o = orc.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xo, quantile=q,
                                                iterations=iters, max_num_clusters=25,
                                                rand_table=[[R] * 64] * B, canonical=True)
    tot_o.sum().backward()
    close(tot_o, tot_r, "convex_loss total", rtol=1e-5)
    close(Xo.grad, Xr.grad, "convex_loss dX", rtol=2e-3, atol=1e-3 * Xr.grad.abs().max().item())
    save("fit_convex_loss", seed=seed, R=R, total=tot_r.detach(), chamfer=ch_r.detach(),
         K=np.array([len(p) for p in prm_r]), labels=torch.stack(lab_r).to(torch.int16),
         dX_norm=Xr.grad.norm(), dX_head=Xr.grad[:, :, :32].contiguous())

    # ---- cuboid variant (--if_cuboid, convex_loss.py:72-76,89): SDF, budget + sampler, chamfer, full loss.
    # trimesh is absent: its box mesh and even surface sampler are replaced, on both sides, by the build's
    # deterministic cuboid_surface(); the reference's own scaling code (src/sample_ellipsoid.py:78-95) runs as is.
    import types

    class _Box:
        def __init__(self):
            self.vertices = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)

    def _sample_even(mesh, count):
        return orc.cuboid_surface(int(count), *np.abs(mesh.vertices).max(0)), None

    SE.trimesh.creation = types.SimpleNamespace(box=lambda extents: _Box())
    SE.trimesh.sample = types.SimpleNamespace(sample_surface_even=_sample_even)
    samples_rc = EU.sample_from_pred_params_cuboid(params_r, 500)
    samples_oc = orc.sample_from_params(params_o, cuboid=True)
    for b in range(B):
        close(samples_oc[b], samples_rc[b], f"cuboid sampled points b={b}", rtol=1e-5, atol=1e-6)
    lc_r = UT.analytic_chamfer_distance(params_r, samples_rc, cham, cuboid=True)
    lc_o, parts_c = orc.analytic_chamfer(params_o, samples_oc, cham, cuboid=True)
    close(lc_o, lc_r, "cuboid analytic chamfer", rtol=1e-5)
    sdf_rc = CL.compute_sdf_cuboid_batch(cham, params_r)
    Xrc = emb.permute(0, 2, 1).clone().requires_grad_(True)
    with patched(torch, "rand", lambda *a, **k: R.clone()), patched(EF, "customsvd", ref_customsvd_canonical), \
            contextlib.redirect_stdout(open(os.devnull, "w")):
        totc_r, chc_r, prmc_r, _ = CL.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xrc, quantile=q,
                                                  iterations=iters, max_num_clusters=25, if_cuboid=True)
    totc_r.sum().backward()
    Xoc = emb.permute(0, 2, 1).clone().requires_grad_(True)
    totc_o, _, _, _ = orc.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xoc, quantile=q, iterations=iters,
                                      max_num_clusters=25, rand_table=[[R] * 64] * B, canonical=True, if_cuboid=True)
    totc_o.sum().backward()
    close(totc_o, totc_r, "cuboid convex_loss total", rtol=1e-5)
    close(Xoc.grad, Xrc.grad, "cuboid convex_loss dX", rtol=2e-3, atol=1e-3 * Xrc.grad.abs().max().item())
    save("fit_cuboid", seed=seed, R=R, loss=lc_r.detach(), dist_st=torch.stack([p[0] for p in parts_c]),
         sdf_ts=torch.stack([p[1] for p in parts_c]), nsamples=np.array([s_.shape[0] for s_ in samples_rc]),
         sdf_head=torch.stack([torch.stack(s_, 1)[:256] for s_ in sdf_rc]).detach(), total=totc_r.detach(),
         samples_head=torch.stack([s_[:64] for s_ in samples_rc]).detach(),
         dX_norm=Xrc.grad.norm(), dX_head=Xrc.grad[:, :, :32].contiguous())

    # ---- optional entropy term (convex_loss.py:209-225) on a fixed quarter of the points
    ent_idx = torch.from_numpy(np.random.default_rng(seed + 9).choice(N, N // 4, replace=False))
    Xe = emb.clone().requires_grad_(True)
    e_r = CL.entropy(torch.nn.functional.normalize(Xe, dim=2)[:, ent_idx])
    e_r.backward()
    Xo2 = emb.clone().requires_grad_(True)
    e_o = orc.entropy(torch.nn.functional.normalize(Xo2, dim=2)[:, ent_idx])
    e_o.backward()
    close(e_o, e_r, "entropy", rtol=1e-6)
    close(Xo2.grad, Xe.grad, "entropy dX", rtol=1e-4, atol=1e-9)
    save("fit_entropy", seed=seed, idx=ent_idx, value=e_r.detach(), dX_head=Xe.grad[:, :64], dX_norm=Xe.grad.norm())

    # ---- known-answer test data: hard weights on analytic ellipsoid surfaces (fitting.py / ellipsoid_fitting_numpy.py:36-45)
    rng = np.random.default_rng(77)
    pts_k, W_k, abc_k, ctr_k = [], [], [], []
    for k in range(3):
        abc = rng.integers(2, 20, size=3).astype(np.float64)
        U, V = orc.fibonacci_uv(500)
        p = np.stack([abc[0] * np.cos(U) * np.sin(V), abc[1] * np.sin(U) * np.sin(V), abc[2] * np.cos(V)], 1)
        th = rng.uniform(0, 2 * np.pi)
        Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        ctr = rng.uniform(0, 1, size=3) * abc.max()
        pts_k.append(p @ Rz + ctr)
        w = np.zeros((500, 3), np.float32)
        w[:, k] = 1
        W_k.append(w)
        abc_k.append(abc)
        ctr_k.append(ctr)
    pk = torch.from_numpy(np.concatenate(pts_k).astype(np.float32))[None]
    wk = torch.from_numpy(np.concatenate(W_k))
    with patched(torch, "rand", lambda *a, **k: R.clone()):
        prm = EF.weighted_ellipsoid_fitting_batch(pk, [wk])
    for k, (r, V, c) in enumerate(prm[0]):
        assert np.allclose(np.sort(r.numpy()), np.sort(abc_k[k]), rtol=2e-2), (r, abc_k[k])
    save("fit_kat", points=pk, W=wk, abc=np.array(abc_k), centres=np.array(ctr_k), R=R,
         r_ref=torch.stack([p[0] for p in prm[0]]), c_ref=torch.stack([p[2] for p in prm[0]]))


def many_cluster_inputs(B=2, N=2048, D=128, seed=61, K=40, M=5000):
    """A cloud of K = 40 tight blobs with K embedding prototypes: more clusters per shape than the loss path's 32 slots."""
    cham, lab = synth.blobs_with_labels(B, M, seed, K=K, sigma=0.05)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    emb = synth.prototype_embedding(lab[:, sel], D, seed + 2, K=K, noise=0.03)   # (0.02: memberships saturate, |dX| ~ 1e-6)
    return torch.from_numpy(cham[:, sel]), torch.from_numpy(cham), torch.from_numpy(emb)


def run_many_clusters(save, eq, close):
    """`--max_num_clusters 49` (args_parser.py:48; the reference's own gaurd_mean_shift accepts 49 clusters,
    src/mean_shift.py:212-226) on shapes with 40 modes: clustering, fit and the whole convex loss with its gradient from the
    reference (same harness conventions as run(): one noise matrix, canonical SVD signs, the Fibonacci sampler), the oracle
    checked against it.  VERDICT r5 item 8: cluster capacity above 32."""
    print("[fit, 40 clusters per shape, max_num_clusters = 49]")
    EF = refshim.ref("src.ellipsoid_fitting")
    EU = refshim.ref("src.ellipsoid_utils")
    CL = refshim.ref("convex_loss")
    SE = refshim.ref("src.sample_ellipsoid")
    B, N, D, seed, q, iters, cap = 2, 2048, 128, 61, 0.01, 10, 49
    pts, cham, emb = many_cluster_inputs(B, N, D, seed)
    R = torch.from_numpy(synth.uniform01((3, 3), seed))

    def ref_customsvd_canonical(M):
        U, S, V = refshim.ref("src.fitting_utils").customsvd(M)
        s = orc.canonical_signs(V).view(1, 3)
        return U * s, S, V * s

    def ref_sample(self, a, b_, c, center, transformation, n=500):
        U, V = orc.fibonacci_uv(int(n))
        p = self.uniform_sample_points_on_ellipsoid(U, V, a, b_, c)
        return p @ transformation.T + center, None

    Ws_r, labs_r = EU.clustering(emb, num_samples=N, quantile=q, iterations=iters, max_num_clusters=cap)
    Ks = [w.shape[1] for w in Ws_r]
    print("  reference: clusters per shape", Ks)
    assert min(Ks) >= 33 and max(Ks) <= cap, Ks
    Ws_o, labs_o, _ = orc.clustering(emb, q, iters, cap)
    for b in range(B):
        assert same_partition(labs_r[b], labs_o[b]), "label partition differs"
    Xr = emb.permute(0, 2, 1).clone().requires_grad_(True)
    with patched(torch, "rand", lambda *a, **k: R.clone()), patched(EF, "customsvd", ref_customsvd_canonical), \
            patched(SE.SampleEllipsoid, "sample", ref_sample):
        tot_r, ch_r, prm_r, lab_r = CL.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xr, quantile=q,
                                                   iterations=iters, max_num_clusters=cap)
    tot_r.sum().backward()
    Xo = emb.permute(0, 2, 1).clone().requires_grad_(True)
    tot_o, ch_o, prm_o, lab_o = orc.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xo, quantile=q, iterations=iters,
                                                max_num_clusters=cap, rand_table=[[R] * 64] * B, canonical=True)
    tot_o.sum().backward()
    close(tot_o, tot_r, "convex_loss total (40 clusters)", rtol=1e-5)
    close(Xo.grad, Xr.grad, "convex_loss dX (40 clusters)", rtol=2e-3, atol=1e-3 * Xr.grad.abs().max().item())
    order = [canonical_order(lab_r[b], Ks[b]) for b in range(B)]
    save("fit_many_clusters", seed=seed, R=R, quantile=q, max_num_clusters=cap, total=tot_r.detach(), chamfer=ch_r.detach(),
         K=np.array([len(p) for p in prm_r]), labels=torch.stack(lab_r).to(torch.int16), dX_norm=Xr.grad.norm(),
         dX_head=Xr.grad[:, :, :32].contiguous(),
         r=torch.stack([torch.stack([prm_r[b][int(k)][0] for k in order[b]] + [torch.zeros(3)] * (cap - Ks[b])) for b in range(B)]).detach(),
         c=torch.stack([torch.stack([prm_r[b][int(k)][2] for k in order[b]] + [torch.zeros(3)] * (cap - Ks[b])) for b in range(B)]).detach())
