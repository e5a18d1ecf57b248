"""CPU: the fitting half of the oracle against the golden vectors captured from the reference
(oracle/make_golden_fit.py): bandwidth, mean-shift iterations + autograd, nms, membership, weighted
ellipsoid fit + autograd, analytic chamfer, convex_loss."""
import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth
from tests_helpers import check_intersection_grads, intersection_case


def _t(a):
    return torch.from_numpy(np.asarray(a))


def fit_inputs(B, N, D, seed, M=5000, noise=0.03):
    cham, lab = synth.blobs_with_labels(B, M, seed)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    return _t(cham[:, sel]), _t(cham), _t(synth.prototype_embedding(lab[:, sel], D, seed + 2, noise=noise))


def same_partition(la, lb):
    pairs = torch.unique(torch.stack([la.long(), lb.long()], 1), dim=0)
    return pairs.shape[0] == torch.unique(la).shape[0] == torch.unique(lb).shape[0]


def test_meanshift_pieces(golden):
    g = golden("fit_meanshift")
    seed = int(g["seed"])
    _, _, emb = fit_inputs(2, 2048, 128, seed)
    G = _t(synth.features(2, 2048, 128, seed + 3))
    b = 0  # one shape keeps the CPU suite fast
    X = emb[b].clone().requires_grad_(True)
    bw = orc.compute_bandwidth(X.detach(), 0.05)
    torch.testing.assert_close(bw, _t(g["bw_0"]), rtol=1e-6, atol=0)
    bw = _t(g["bw_0"])  # the fixture's gradients were taken at the reference's bandwidth
    Z = orc.mean_shift_iterations(X, bw, 10)
    (Z * G[b]).sum().backward()
    torch.testing.assert_close(Z[:64].detach(), _t(g["Z_head_0"]), rtol=1e-5, atol=1e-6)
    ref = _t(g["dX_head_0"])
    torch.testing.assert_close(X.grad[:64], ref, rtol=1e-3, atol=1e-4 * ref.abs().max().item())
    _, ids, labels = orc.nms(Z.detach(), Z.detach(), bw)
    assert ids.shape[0] == int(g["K_0"]) and same_partition(labels, _t(g["labels_0"]).long())


def test_fit_and_chamfer(golden):
    ge, gc = golden("fit_ellipsoid"), golden("fit_chamfer")
    seed = int(ge["seed"])
    pts, cham, _ = fit_inputs(2, 2048, 128, seed)
    R = _t(ge["R"])
    Ws = [_t(ge[f"W_{b}"]).requires_grad_(True) for b in range(2)]
    params = orc.fit_ellipsoids_batch(pts, Ws, [[R] * w.shape[1] for w in Ws], canonical=True)
    gr = _t(ge["grad_seed_table"])
    loss = 0
    for b in range(2):
        for k, (r, V, c) in enumerate(params[b]):
            torch.testing.assert_close(r.detach(), _t(ge[f"r_{b}"])[k], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(V.detach(), _t(ge[f"V_{b}"])[k], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(c.detach(), _t(ge[f"c_{b}"])[k], rtol=1e-5, atol=1e-6)
            loss = loss + (r * gr[k, 0:3]).sum() + (V * gr[k, 3:12].view(3, 3)).sum() + (c * gr[k, 12:15]).sum()
    loss.backward()
    for b in range(2):
        ref = _t(ge[f"dW_{b}"])
        torch.testing.assert_close(Ws[b].grad, ref, rtol=1e-3, atol=1e-5 * ref.abs().max().item())
    samples = orc.sample_from_params(params)
    assert [s.shape[0] for s in samples] == list(gc["nsamples"])
    l, parts = orc.analytic_chamfer(params, samples, cham)
    torch.testing.assert_close(l.detach(), _t(gc["loss"]), rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(torch.stack([p[0] for p in parts]), _t(gc["dist_st"]), rtol=1e-5, atol=1e-9)


def test_cuboid_variant(golden):
    """--if_cuboid (convex_loss.py:72-76,89): cuboid SDF, area budget, box-surface samples, analytic chamfer."""
    ge, gc = golden("fit_ellipsoid"), golden("fit_cuboid")
    pts, cham, _ = fit_inputs(2, 2048, 128, int(ge["seed"]))
    params = [[(_t(ge[f"r_{b}"])[k], _t(ge[f"V_{b}"])[k], _t(ge[f"c_{b}"])[k]) for k in range(ge[f"r_{b}"].shape[0])]
              for b in range(2)]
    samples = orc.sample_from_params(params, cuboid=True)
    assert [s.shape[0] for s in samples] == list(gc["nsamples"])
    for b in range(2):
        torch.testing.assert_close(samples[b][:64], _t(gc["samples_head"])[b], rtol=1e-5, atol=1e-6)
        sdf = torch.stack([orc.sdf_cuboid(cham[b][:256], c, r, V) for r, V, c in params[b]], 1)
        torch.testing.assert_close(sdf, _t(gc["sdf_head"])[b], rtol=1e-5, atol=1e-6)
    l, parts = orc.analytic_chamfer(params, samples, cham, cuboid=True)
    torch.testing.assert_close(l, _t(gc["loss"]), rtol=1e-5, atol=1e-9)
    # the box-surface table: on the faces, area-proportional, evenly spread
    p = orc.cuboid_surface(6000, 1.0, 2.0, 3.0) / np.array([[1.0, 2.0, 3.0]])
    assert np.allclose(np.abs(p).max(1), 1.0) and abs(p.mean(0)).max() < 2e-2
    on_z = (np.abs(p[:, 2]) == 1.0).mean()
    assert abs(on_z - 2.0 / 11.0) < 2e-3     # area share of the two z faces: ab / (ab + bc + ca) = 2 / 11


def test_known_answer(golden):
    g = golden("fit_kat")
    prm = orc.fit_ellipsoids_batch(_t(g["points"]), [_t(g["W"])], [[_t(g["R"])] * 3])
    for k, (r, V, c) in enumerate(prm[0]):
        assert np.allclose(np.sort(r.numpy()), np.sort(g["abc"][k]), rtol=2e-2)
        torch.testing.assert_close(r, _t(g["r_ref"])[k], rtol=1e-5, atol=1e-5)


def test_sample_budget_and_table():
    U, V = orc.fibonacci_uv(1000)
    assert U.shape == (1000,) and float(V.min()) > 0 and float(V.max()) < np.pi
    p = torch.stack([torch.cos(U) * torch.sin(V), torch.sin(U) * torch.sin(V), torch.cos(V)], 1)
    assert p.mean(0).abs().max() < 5e-3  # evenly spread over the sphere
    params = [(torch.tensor([1.0, 1.0, 1.0]), torch.eye(3), torch.zeros(3)), (torch.tensor([2.0, 2.0, 2.0]), torch.eye(3), torch.zeros(3))]
    n = orc.sample_budget(params)
    assert n.sum() == 10000 and abs(n[1] / n[0] - 4.0) < 0.01


def test_subsampled_bandwidth_matches_reference(golden):
    """src/mean_shift.py:148-151 with num_samples = 1000 < N = 2048 (the default of clustering(X), fitting.py:43): the
    oracle on the reference's own row subset."""
    from tests_helpers import fit_inputs
    g = golden("fit_bandwidth_sub")
    _, _, emb = fit_inputs(2, 2048, 128, int(g["seed"]))
    for b in range(2):
        bw = orc.compute_bandwidth(emb[b], 0.05, rows=g["rows_%d" % b].astype(np.int64))
        assert abs(float(bw) - float(g["bw_%d" % b])) <= 1e-6 * float(g["bw_%d" % b])


def test_bandwidth_with_more_samples_than_rows_matches_reference(golden):
    """src/mean_shift.py:151-155 with num_samples = 1000 > N = 512 (clustering(X)'s default on a small cloud): all N rows,
    K = int(quantile * num_samples) = 50 -- not int(quantile * N) = 25."""
    from tests_helpers import fit_inputs
    g = golden("fit_bandwidth_over")
    _, _, emb = fit_inputs(2, int(g["N"]), 128, int(g["seed"]))
    for b in range(2):
        bw = orc.compute_bandwidth(emb[b], float(g["quantile"]), num_samples=int(g["num_samples"]))
        assert abs(float(bw) - float(g["bw_%d" % b])) <= 1e-6 * float(g["bw_%d" % b])


def test_mean_shift_variants_match_reference(golden):
    """The epanechnikov kernel and mean_shift_eff_ (src/mean_shift.py:70-74, :86-136; unused by the loss) of the oracle against
    values captured from the reference."""
    from tests_helpers import fit_inputs
    from prifit_amd import synth
    g = golden("fit_meanshift_variants")
    seed, N, D = int(g["seed"]), 512, 32
    _, _, emb = fit_inputs(1, N, D, seed, M=1000, noise=0.1)
    X0 = emb[0]
    G = torch.from_numpy(synth.features(1, N, D, seed + 1))[0]
    rows = torch.from_numpy(g["rows"].astype(np.int64))
    b = torch.tensor(0.9)
    for name, fn, gg in (("epa", lambda X: orc.mean_shift_iterations(X, b, 3, "epa"), G),
                         ("eff", lambda X: orc.mean_shift_eff(X, X[rows], b, 3), G[: N // 2]),
                         ("eff_epa", lambda X: orc.mean_shift_eff(X, X[rows], b, 3, "epa"), G[: N // 2])):
        X = X0.clone().requires_grad_(True)
        Z = fn(X)
        (Z * gg).sum().backward()
        assert torch.allclose(Z[:64].detach(), torch.from_numpy(g["Z_" + name]), rtol=1e-5, atol=1e-6)
        assert torch.allclose(X.grad[:64], torch.from_numpy(g["dX_" + name]), rtol=1e-4, atol=1e-6 * float(X.grad.abs().max()))


def test_nms_with_distinct_centres_and_epanechnikov_guard_match_reference(golden):
    """src/mean_shift.py:162-202 called as nms(shifted points, original points, b), and src/ellipsoid_utils.py:9-27 with
    kernel_type forwarded (epanechnikov kernel, src/mean_shift.py:70-74, with quantile-doubling retries)."""
    import tests_helpers as H
    g = golden("fit_nms_pair")
    seed, N, D = int(g["seed"]), 512, 32
    _, _, emb = H.fit_inputs(2, N, D, seed, M=1000, noise=0.1)
    for b in range(2):
        X = emb[b]
        bw = orc.compute_bandwidth(X, 0.05)
        assert abs(float(bw) - float(g["bw_%d" % b])) <= 1e-6 * float(bw)
        Z = orc.mean_shift_iterations(X, bw, 4)
        _, ids, labels = orc.nms(Z, X, bw)
        assert torch.equal(ids, _t(g["ids_%d" % b]).long()) and torch.equal(labels, _t(g["labels_%d" % b]).long())
    centers, bw, labels, ids, Z, q = orc.guard_mean_shift(emb[0], float(g["q0"]), int(g["iters"]), int(g["cap"]), kernel_type="epa")
    assert abs(q - float(g["epa_quantiles"][-1])) < 1e-12 and centers.shape[0] == int(g["epa_K"])
    assert abs(float(bw) - float(g["epa_bw"])) <= 1e-6 * float(bw)
    assert torch.equal(labels, _t(g["epa_labels"]).long())


def test_center_gather_gradient_golden(golden):
    """d/dX sum(G * new_X[ids]) -- the only differentiable use of the shifted points (src/mean_shift.py:44-46) -- of the
    oracle against the reference's autograd (fit_center_grad.npz)."""
    g = golden("fit_center_grad")
    seed = int(g["seed"])
    _, _, emb = fit_inputs(2, 2048, 128, seed)
    b = 1
    X = emb[b].clone().requires_grad_(True)
    ids = _t(g[f"ids_{b}"]).long()
    Z = orc.mean_shift_iterations(X, _t(g[f"bw_{b}"]), int(g["iterations"]))
    (Z[ids] * _t(g[f"G_{b}"])).sum().backward()
    torch.testing.assert_close(Z[ids].detach(), _t(g[f"centres_{b}"]), rtol=1e-5, atol=1e-6)
    ref = _t(g[f"dX_ids_{b}"])
    torch.testing.assert_close(X.grad[ids], ref, rtol=1e-3, atol=1e-4 * ref.abs().max().item())
    assert abs(float(X.grad.norm()) - float(g[f"dX_norm_{b}"])) <= 1e-3 * float(g[f"dX_norm_{b}"])


def test_prune_points_golden(golden):
    """prune_points (convex_loss.py:444-470): the oracle keeps exactly the points the reference kept (fit_prune.npz)."""
    g = golden("fit_prune")
    params = [[(_t(g[f"r_{b}"][k]), _t(g[f"V_{b}"][k]), _t(g[f"c_{b}"][k])) for k in range(g[f"r_{b}"].shape[0])]
              for b in range(3)]
    pts = orc.sample_from_params(params)
    kept = orc.prune_points(pts, params)
    for b in range(3):
        assert pts[b].shape[0] == int(g[f"n_{b}"])
        assert kept[b].shape[0] == int(g[f"keep_{b}"].sum())
        assert torch.equal(kept[b], pts[b][_t(g[f"keep_{b}"])])
        torch.testing.assert_close(kept[b].sum(0), _t(g[f"kept_sum_{b}"]), rtol=1e-5, atol=1e-4)


INTERSECTION_VARIANTS = {
    "surface": lambda P, surf, pts: orc.intersection_loss_surface(P, surf),
    "surface_cuboid": lambda P, surf, pts: orc.intersection_loss_surface(P, surf, cuboid=True),
    "volume": lambda P, surf, pts: orc.intersection_loss_volume(P, surf),
    "volume_2": lambda P, surf, pts: orc.intersection_loss_volume_2(P, pts),
    "volume_4": lambda P, surf, pts: orc.intersection_loss_volume_4(P, pts),
}


@pytest.mark.parametrize("name", sorted(INTERSECTION_VARIANTS))
def test_unused_intersection_variants_golden(golden, name):
    """convex_loss.py:106, :163, :227, :346, :416 (kept upstream, never called): oracle value and gradient with respect to
    every (r, V, c) against the reference's autograd (fit_intersections.npz)."""
    g = golden("fit_intersections")
    P, surf, pts = intersection_case(g)
    loss = INTERSECTION_VARIANTS[name](P, surf, pts)
    torch.testing.assert_close(loss.detach().reshape(()), _t(g[f"{name}_loss"]), rtol=1e-5, atol=1e-9)
    loss.backward()
    check_intersection_grads(g, name, P)


def test_sample_axis_golden(golden):
    g = golden("fit_intersections")
    got = orc.sample_axis(_t(g["r_0"][1]), _t(g["V_0"][1]), _t(g["c_0"][1]))
    torch.testing.assert_close(got, _t(g["axis_samples"]), rtol=1e-6, atol=1e-7)


def test_many_clusters_golden(golden):
    """--max_num_clusters 49 (args_parser.py:48; src/mean_shift.py:212-226) on shapes with 40 modes: the oracle's clustering,
    fit, loss and its gradient against the reference's (fit_many_clusters.npz)."""
    from tests_helpers import many_cluster_inputs
    g = golden("fit_many_clusters")
    pts, cham, emb = many_cluster_inputs(seed=int(g["seed"]))
    R = _t(g["R"])
    X = emb.permute(0, 2, 1).clone().requires_grad_(True)
    total, ch, params, labels = orc.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), X, quantile=float(g["quantile"]),
                                                iterations=10, max_num_clusters=int(g["max_num_clusters"]),
                                                rand_table=[[R] * 64] * 2, canonical=True)
    total.sum().backward()
    assert [len(p) for p in params] == list(g["K"]) and min(g["K"]) >= 33
    for b in range(2):
        assert same_partition(labels[b], _t(g["labels"])[b].long())
    torch.testing.assert_close(total.detach(), _t(g["total"]), rtol=1e-5, atol=1e-8)
    ref = _t(g["dX_head"])
    torch.testing.assert_close(X.grad[:, :, :32], ref, rtol=2e-3, atol=1e-3 * ref.abs().max().item())
