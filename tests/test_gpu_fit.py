"""GPU parity of the fitting path (through the C ABI) vs the oracle and the goldens captured from the
reference: bandwidth, mean-shift iterations (+autograd), nms, membership, weighted ellipsoid fit
(+autograd), ellipsoid SDF, surface sampling + nearest target, analytic chamfer, convex_loss.

Tolerances: fp32 fitting residuals 1e-4 (BASELINE.json north_star); cluster ids / order are rounding
noise in the reference itself, so clusters are compared through the label partition."""
import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def fit_inputs(B, N, D, seed, M=5000, noise=0.03):
    cham, lab = synth.blobs_with_labels(B, M, seed)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    return _t(cham[:, sel]), _t(cham), _t(synth.prototype_embedding(lab[:, sel], D, seed + 2, noise=noise))


def same_partition(la, lb):
    pairs = torch.unique(torch.stack([la.long(), lb.long()], 1), dim=0)
    return pairs.shape[0] == torch.unique(la).shape[0] == torch.unique(lb).shape[0]


@pytest.fixture(scope="module")
def F(hiplib):
    assert torch.cuda.is_available()
    from prifit_amd import fit_ops
    return fit_ops


@pytest.mark.parametrize("B,N,D", [(3, 2048, 128), (2, 256, 64), (1, 128, 32)])
def test_chord_matrix_symmetric_kernel(F, B, N, D):
    """chord_matrix(X, X) through prifit_chord_sym_f32 (upper triangle of tiles + transposed writes) against the general
    batched GEMM with the chord epilogue: the same bits."""
    X = torch.nn.functional.normalize(torch.randn(B, N, D, generator=torch.Generator().manual_seed(5)), dim=2).cuda()
    old = F.CHORD_SYM
    try:
        F.CHORD_SYM = True
        a = F.chord_matrix(X, X)
        F.CHORD_SYM = False
        b = F.chord_matrix(X, X)
    finally:
        F.CHORD_SYM = old
    assert torch.equal(a, b)
    ref = 2 - 2 * X.double() @ X.double().transpose(1, 2)
    assert (a.double() - ref).abs().max() < 1e-5


def test_kth_smallest_and_bandwidth(F, golden):
    from prifit_amd._lib import call, cur_stream, ptr
    import ctypes
    for C, k in ((2048, 102), (1000, 1), (300, 300), (4096, 77)):
        M = torch.randn(37, C, generator=torch.Generator().manual_seed(C))
        M[:, 0:C // 2:5] = M[:, 1:C // 2:5][:, : M[:, 0:C // 2:5].shape[1]]  # exact duplicates
        out = torch.empty(37, device="cuda")
        call("prifit_kth_smallest_rows", ptr(M.cuda()), ctypes.c_longlong(37), C, k, ptr(out), cur_stream())
        assert torch.equal(out.cpu(), torch.topk(M, k, dim=1, largest=False)[0][:, -1])
    g = golden("fit_meanshift")
    _, _, emb = fit_inputs(2, 2048, 128, int(g["seed"]))
    bw = F.compute_bandwidth(emb.cuda(), 0.05)
    torch.testing.assert_close(bw.cpu(), torch.stack([_t(g["bw_0"]), _t(g["bw_1"])]), rtol=1e-5, atol=0)


def test_mean_shift_iterations_fwd_bwd(F, golden):
    g = golden("fit_meanshift")
    seed = int(g["seed"])
    _, _, emb = fit_inputs(2, 2048, 128, seed)
    G = _t(synth.features(2, 2048, 128, seed + 3))
    bw = torch.stack([_t(g["bw_0"]), _t(g["bw_1"])])
    X = emb.cuda().requires_grad_(True)
    Z = F.MeanShiftFn.apply(X, bw.cuda(), 10)
    (Z * G.cuda()).sum().backward()
    for b in range(2):
        torch.testing.assert_close(Z[b, :64].detach().cpu(), _t(g[f"Z_head_{b}"]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(Z[b].detach().sum(0).cpu(), _t(g[f"Z_colsum_{b}"]), rtol=1e-4, atol=1e-3)
        ref = _t(g[f"dX_head_{b}"])
        torch.testing.assert_close(X.grad[b, :64].cpu(), ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())
        assert abs(X.grad[b].norm().item() - float(g[f"dX_norm_{b}"])) < 2e-3 * float(g[f"dX_norm_{b}"])
    # small ragged case against the oracle directly (N not a multiple of the tiles, D = 32 as in fitting.py)
    Xs = torch.nn.functional.normalize(_t(synth.features(1, 300, 32, 5))[0], dim=1)
    bws = orc.compute_bandwidth(Xs, 0.1)
    Xo = Xs.clone().requires_grad_(True)
    Zo = orc.mean_shift_iterations(Xo, bws, 4)
    Gs = _t(synth.features(1, 300, 32, 6))[0]
    (Zo * Gs).sum().backward()
    Xg = Xs.cuda().unsqueeze(0).requires_grad_(True)
    Zg = F.MeanShiftFn.apply(Xg, bws.reshape(1).cuda(), 4)
    (Zg[0] * Gs.cuda()).sum().backward()
    torch.testing.assert_close(Zg[0].detach().cpu(), Zo.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(Xg.grad[0].cpu(), Xo.grad, rtol=2e-3, atol=2e-4 * Xo.grad.abs().max().item())


@pytest.mark.parametrize("N,mode", [(200, "gemm"), (200, "fused"), (333, "gemm"), (2048, "fused"), (2048, "hybrid"), (200, "hybrid")])
def test_mean_shift_fused_ragged(F, N, mode, monkeypatch):
    monkeypatch.setattr(F, "BWD_MODE", mode)
    """D = 128 takes the flash-style fused kernels; N not a multiple of the 64-row tiles (and N % 4 != 0, which
    falls back to the GEMM chain on K^T for the backward pass)."""
    Xs = torch.nn.functional.normalize(_t(synth.features(2, N, 128, 15)), dim=-1)
    bws = torch.stack([orc.compute_bandwidth(Xs[b], 0.1) for b in range(2)])
    Gs = _t(synth.features(2, N, 128, 16))
    Xo = Xs.clone().requires_grad_(True)
    Zo = torch.stack([orc.mean_shift_iterations(Xo[b], bws[b], 3) for b in range(2)])
    (Zo * Gs).sum().backward()
    Xg = Xs.cuda().requires_grad_(True)
    Zg = F.MeanShiftFn.apply(Xg, bws.cuda(), 3)
    (Zg * Gs.cuda()).sum().backward()
    torch.testing.assert_close(Zg.detach().cpu(), Zo.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(Xg.grad.cpu(), Xo.grad, rtol=2e-3, atol=2e-4 * Xo.grad.abs().max().item())


def test_nms_and_membership(F, golden):
    g = golden("fit_meanshift")
    seed = int(g["seed"])
    _, _, emb = fit_inputs(2, 2048, 128, seed)
    bw = torch.stack([_t(g["bw_0"]), _t(g["bw_1"])])
    Zo = torch.stack([orc.mean_shift_iterations(emb[b], bw[b], 10) for b in range(2)])
    ids, count, labels, used = F.nms(Zo.cuda(), bw.cuda())
    for b in range(2):
        K = int(count[b])
        assert K == int(g[f"K_{b}"])
        assert same_partition(labels[b].cpu(), _t(g[f"labels_{b}"]).long())
        assert int(used[b].sum()) == K
        # membership with the HIP-chosen centres vs the oracle on the same centres
        cen = Zo[b][ids[b, :K].long().cpu()]
        Wo = orc.membership(cen, emb[b], bw[b])
        from prifit_amd.src.mean_shift import MeanShift
        Wg = MeanShift().membership(cen.cuda(), emb[b].cuda(), bw[b].cuda())
        torch.testing.assert_close(Wg.cpu(), Wo, rtol=1e-4, atol=1e-6)
        # column sums in partition-canonical order vs the reference's
        first = [int(torch.nonzero(labels[b].cpu() == k).min()) for k in range(K)]
        order = torch.tensor(np.argsort(first, kind="stable"))
        torch.testing.assert_close(Wg.cpu().t()[:, order].sum(0), _t(g[f"W_colsum_{b}"]), rtol=1e-4, atol=1e-2)


def test_membership_backward(F):
    Bt, N, D, K = 2, 500, 32, 5
    X = torch.nn.functional.normalize(_t(synth.features(Bt, N, D, 1)), dim=-1)
    cen = torch.nn.functional.normalize(_t(synth.features(Bt, K, D, 2)), dim=-1)
    bw = torch.tensor([0.7, 0.4])
    gW = _t(synth.features(Bt, N, K, 3))
    Xo, co = X.clone().requires_grad_(True), cen.clone().requires_grad_(True)
    loss = sum((orc.membership(co[b], Xo[b], bw[b]).t() * gW[b]).sum() for b in range(Bt))
    loss.backward()
    cpad = torch.zeros(Bt, F.KM, D)
    cpad[:, :K] = cen
    Xg, cg = X.cuda().requires_grad_(True), cpad.cuda().requires_grad_(True)
    W = F.MembershipFn.apply(cg, Xg, bw.cuda(), torch.tensor([K, K], dtype=torch.int32, device="cuda"))
    gpad = torch.zeros(Bt, N, F.KM)
    gpad[:, :, :K] = gW
    (W * gpad.cuda()).sum().backward()
    torch.testing.assert_close(Xg.grad.cpu(), Xo.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(cg.grad[:, :K].cpu(), co.grad, rtol=1e-3, atol=1e-4)
    assert cg.grad[:, K:].abs().max().item() == 0


def test_ellipsoid_fit_fwd_bwd(F, golden):
    g = golden("fit_ellipsoid")
    seed = int(g["seed"])
    pts, _, _ = fit_inputs(2, 2048, 128, seed)
    R = _t(g["R"]).cuda()
    gr = _t(g["grad_seed_table"])
    from prifit_amd.src.ellipsoid_fitting import weighted_ellipsoid_fitting_batch
    Ws = [_t(g[f"W_{b}"]).cuda().requires_grad_(True) for b in range(2)]
    params = weighted_ellipsoid_fitting_batch(pts.cuda(), Ws, rand_table=R, canonical=True)
    loss = 0
    for b in range(2):
        assert len(params[b]) == g[f"r_{b}"].shape[0]
        for k, (r, V, c) in enumerate(params[b]):
            torch.testing.assert_close(r.detach().cpu(), _t(g[f"r_{b}"])[k], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(V.detach().cpu(), _t(g[f"V_{b}"])[k], rtol=1e-3, atol=1e-4)
            torch.testing.assert_close(c.detach().cpu(), _t(g[f"c_{b}"])[k], rtol=1e-4, atol=1e-5)
            gk = gr[k].cuda()
            loss = loss + (r * gk[0:3]).sum() + (V * gk[3:12].view(3, 3)).sum() + (c * gk[12:15]).sum()
    loss.backward()
    for b in range(2):
        ref = _t(g[f"dW_{b}"])
        torch.testing.assert_close(Ws[b].grad.cpu(), ref, rtol=2e-3, atol=1e-4 * ref.abs().max().item())


def test_fit_known_answer(F, golden):
    """hard one-hot weights on analytic ellipsoid surfaces: semi-axes are recovered (fitting.py, ellipsoid_fitting_numpy.py:36-45)."""
    g = golden("fit_kat")
    from prifit_amd.src.ellipsoid_fitting import weighted_ellipsoid_fitting_batch
    prm = weighted_ellipsoid_fitting_batch(_t(g["points"]).cuda(), [_t(g["W"]).cuda()], rand_table=_t(g["R"]).cuda())
    assert len(prm[0]) == 3
    for k, (r, V, c) in enumerate(prm[0]):
        assert np.allclose(np.sort(r.cpu().numpy()), np.sort(g["abc"][k]), rtol=2e-2)
        torch.testing.assert_close(r.cpu(), _t(g["r_ref"])[k], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(c.cpu(), _t(g["c_ref"])[k], rtol=1e-4, atol=1e-4)
        assert abs(abs(torch.det(V).item()) - 1) < 1e-4 and torch.det(V).item() > 0


def test_sdf_and_sample_chamfer(F, golden):
    ge, gc = golden("fit_ellipsoid"), golden("fit_chamfer")
    seed = int(ge["seed"])
    pts, cham, _ = fit_inputs(2, 2048, 128, seed)
    K = ge["r_0"].shape[0]
    r = torch.zeros(2, F.KM, 3); V = torch.eye(3).repeat(2, F.KM, 1, 1); c = torch.zeros(2, F.KM, 3)
    valid = torch.zeros(2, F.KM, dtype=torch.int32)
    for b in range(2):
        r[b, :K], V[b, :K], c[b, :K], valid[b, :K] = _t(ge[f"r_{b}"]), _t(ge[f"V_{b}"]), _t(ge[f"c_{b}"]), 1
    rg, Vg, cg = (t.cuda().requires_grad_(True) for t in (r, V, c))
    from prifit_amd.convex_loss import analytic_chamfer_distance
    loss, (dist_st, sdf_ts) = analytic_chamfer_distance(rg, Vg, cg, valid.cuda(), cham.cuda())
    torch.testing.assert_close(sdf_ts.detach().cpu(), _t(gc["sdf_ts"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dist_st.detach().cpu(), _t(gc["dist_st"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(loss.detach().cpu(), _t(gc["loss"]), rtol=1e-4, atol=1e-8)
    loss.backward()
    # gradients vs the oracle's autograd on the same parameters
    ro, Vo, co = r.clone().requires_grad_(True), V.clone().requires_grad_(True), c.clone().requires_grad_(True)
    params = [[(ro[b, k], Vo[b, k], co[b, k]) for k in range(K)] for b in range(2)]
    lo, _ = orc.analytic_chamfer(params, orc.sample_from_params(params), cham)
    lo.backward()
    for got, ref in ((rg.grad, ro.grad), (Vg.grad, Vo.grad), (cg.grad, co.grad)):
        torch.testing.assert_close(got.cpu(), ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())


def test_cuboid_variant(F, golden):
    """--if_cuboid: cuboid SDF / budget / box-surface samples / chamfer vs the reference fixture, gradients vs the
    oracle's autograd, and the full convex_loss(if_cuboid=True) with its embedding gradient."""
    ge, gc = golden("fit_ellipsoid"), golden("fit_cuboid")
    seed = int(ge["seed"])
    pts, cham, emb = fit_inputs(2, 2048, 128, seed)
    K = ge["r_0"].shape[0]
    r = torch.zeros(2, F.KM, 3); V = torch.eye(3).repeat(2, F.KM, 1, 1); c = torch.zeros(2, F.KM, 3)
    valid = torch.zeros(2, F.KM, dtype=torch.int32)
    for b in range(2):
        r[b, :K], V[b, :K], c[b, :K], valid[b, :K] = _t(ge[f"r_{b}"]), _t(ge[f"V_{b}"]), _t(ge[f"c_{b}"]), 1
    rg, Vg, cg = (t.cuda().requires_grad_(True) for t in (r, V, c))
    from prifit_amd.convex_loss import SdfMatrixFn, analytic_chamfer_distance, convex_loss
    sdf = SdfMatrixFn.apply(cham.cuda(), rg, Vg, cg, valid.cuda(), True)
    torch.testing.assert_close(sdf[:, :256, :K].detach().cpu(), _t(gc["sdf_head"]), rtol=1e-4, atol=1e-6)
    loss, (dist_st, sdf_ts) = analytic_chamfer_distance(rg, Vg, cg, valid.cuda(), cham.cuda(), cuboid=True)
    torch.testing.assert_close(sdf_ts.detach().cpu(), _t(gc["sdf_ts"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dist_st.detach().cpu(), _t(gc["dist_st"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(loss.detach().cpu(), _t(gc["loss"]), rtol=1e-4, atol=1e-8)
    gm = _t(synth.features(2, 5000, F.KM, 41)) * 1e-3
    (loss + (sdf * gm.cuda()).sum()).backward()
    ro, Vo, co = r.clone().requires_grad_(True), V.clone().requires_grad_(True), c.clone().requires_grad_(True)
    params = [[(ro[b, k], Vo[b, k], co[b, k]) for k in range(K)] for b in range(2)]
    lo, _ = orc.analytic_chamfer(params, orc.sample_from_params(params, cuboid=True), cham, cuboid=True)
    for b in range(2):
        so = torch.stack([orc.sdf_cuboid(cham[b], co[b, k], ro[b, k], Vo[b, k]) for k in range(K)], 1)
        lo = lo + (so * gm[b, :, :K]).sum()
    lo.backward()
    for got, ref in ((rg.grad, ro.grad), (Vg.grad, Vo.grad), (cg.grad, co.grad)):
        torch.testing.assert_close(got.cpu(), ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())
    X = emb.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    total, l, _, _ = convex_loss(pts.permute(0, 2, 1).cuda(), cham.permute(0, 2, 1).cuda(), X, quantile=0.05,
                                 iterations=10, max_num_clusters=25, rand_table=_t(gc["R"]).cuda(), canonical=True,
                                 if_cuboid=True)
    total.sum().backward()
    torch.testing.assert_close(total.detach().cpu(), _t(gc["total"]), rtol=1e-4, atol=1e-7)
    ref = _t(gc["dX_head"])
    assert abs(X.grad.norm().item() - float(gc["dX_norm"])) < 1e-2 * float(gc["dX_norm"])
    torch.testing.assert_close(X.grad[:, :, :32].cpu(), ref, rtol=1e-2, atol=1e-3 * ref.abs().max().item())


@pytest.mark.parametrize("split", ["0", "bf16x6", "fp16x3"])
def test_convex_loss_end_to_end(F, golden, split, monkeypatch):
    """split != "0": the labelled experiment of csrc/meanshift_split.hip (mean-shift forward products on the 16-bit matrix
    pipe, error-compensated) must pass at the SAME tolerances; "bf16x3" (operands to 2^-16) is not expected to and is
    measured in test_split_mean_shift_products_error_against_fp64 only."""
    monkeypatch.setattr(F, "MS_SPLIT", split)
    assert split == "0" or F.split_mode(2048, 128) > 0
    launches0 = F.split_launches
    g = golden("fit_convex_loss")
    seed = int(g["seed"])
    pts, cham, emb = fit_inputs(2, 2048, 128, seed)
    from prifit_amd.convex_loss import convex_loss
    X = emb.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    total, l, params, labels, info = convex_loss(pts.permute(0, 2, 1).cuda(), cham.permute(0, 2, 1).cuda(), X,
                                                 quantile=0.05, iterations=10, max_num_clusters=25,
                                                 rand_table=_t(g["R"]).cuda(), canonical=True, return_info=True)
    total.sum().backward()
    assert F.split_launches - launches0 == (10 if split != "0" else 0)   # the experiment's kernel did (not) run
    assert total.shape == (1, 1) and l.shape == (1, 1)
    assert [len(p) for p in params] == list(g["K"])
    for b in range(2):
        assert same_partition(labels[b].cpu(), _t(g["labels"])[b].long())
    torch.testing.assert_close(total.detach().cpu(), _t(g["total"]), rtol=1e-4, atol=1e-7)
    ref = _t(g["dX_head"])
    assert abs(X.grad.norm().item() - float(g["dX_norm"])) < 1e-2 * float(g["dX_norm"])
    torch.testing.assert_close(X.grad[:, :, :32].cpu(), ref, rtol=1e-2, atol=1e-3 * ref.abs().max().item())


def test_convex_loss_with_forty_clusters_per_shape(F, golden):
    """Cluster capacity above the loss path's 32 slots: `max_num_clusters = 49` -- the reference's own
    gaurd_mean_shift accepts 49, src/mean_shift.py:212-226 -- on shapes with 40 modes takes 64 slots per shape (fit_ops.slots_for):
    row-sparse mean-shift backward with 40 live rows, membership / fit / SDF / sampling kernels at K = 64.  Clusters, partition,
    loss at 1e-4 and the embedding gradient against the REFERENCE (fit_many_clusters.npz); 25 clusters still take 32 slots."""
    from tests_helpers import many_cluster_inputs
    from prifit_amd.convex_loss import convex_loss
    g = golden("fit_many_clusters")
    assert F.slots_for(25) == 32 and F.slots_for(32) == 32 and F.slots_for(33) == 64 and F.slots_for(49) == 64
    with pytest.raises(ValueError):
        F.slots_for(65)
    pts, cham, emb = many_cluster_inputs(seed=int(g["seed"]))
    X = emb.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    total, l, params, labels, info = convex_loss(pts.permute(0, 2, 1).cuda(), cham.permute(0, 2, 1).cuda(), X,
                                                 quantile=float(g["quantile"]), iterations=10, max_num_clusters=int(g["max_num_clusters"]),
                                                 rand_table=_t(g["R"]).cuda(), canonical=True, return_info=True)
    total.sum().backward()
    assert [len(p) for p in params] == list(g["K"]) and min(g["K"]) >= 33
    for b in range(2):
        assert same_partition(labels[b].cpu(), _t(g["labels"])[b].long())
    torch.testing.assert_close(total.detach().cpu(), _t(g["total"]), rtol=1e-4, atol=1e-7)
    ref = _t(g["dX_head"])
    assert abs(X.grad.norm().item() - float(g["dX_norm"])) < 2e-2 * float(g["dX_norm"])
    torch.testing.assert_close(X.grad[:, :, :32].cpu(), ref, rtol=2e-2, atol=2e-3 * ref.abs().max().item())
    # the fitted ellipsoids themselves, in partition-canonical order (ascending smallest member index)
    for b in range(2):
        K = int(g["K"][b])
        lab = labels[b].cpu()
        first = [int(torch.nonzero(lab == k).min()) for k in range(K)]
        order = np.argsort(np.array(first), kind="stable")
        r = torch.stack([params[b][int(k)][0] for k in order]).detach().cpu()
        c = torch.stack([params[b][int(k)][2] for k in order]).detach().cpu()
        torch.testing.assert_close(c, _t(g["c"])[b, :K], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(r, _t(g["r"])[b, :K], rtol=1e-3, atol=1e-4)


def test_cluster_retry_and_degenerate(F):
    """quantile-doubling retry (src/ellipsoid_utils.py:19-27) and a shape whose fit is ill-conditioned."""
    B, N, D = 2, 1024, 32
    rng = np.random.default_rng(3)
    # 40 tight prototypes -> more than 25 clusters at a small quantile -> the loop must enlarge the bandwidth
    proto = rng.normal(size=(40, D)); proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    lab = rng.integers(0, 40, size=(B, N))
    emb = proto[lab] + 0.003 * rng.normal(size=(B, N, D))
    emb = _t((emb / np.linalg.norm(emb, axis=-1, keepdims=True)).astype(np.float32))
    cl = F.cluster(emb.cuda(), 0.01, 5, 25)
    Ws, labs, info = orc.clustering(emb, 0.01, 5, 25)
    for b in range(B):
        assert int(cl["count"][b]) == Ws[b].shape[1] <= 25
        assert abs(cl["quantile"][b] - info[b]["quantile"]) < 1e-12
        assert same_partition(cl["labels"][b].cpu(), labs[b])
    # coplanar points: smallest singular value ~ 0 -> cond > 1e5 -> dropped (valid = 0), loss stays finite
    pts = _t(synth.cloud("cube", 1, 512, 1)); pts[..., 2] = 0.0
    W = torch.zeros(1, 512, F.KM); W[:, :, 0] = 1.0
    r, V, c, valid = F.EllipsoidFitFn.apply(pts.cuda(), W.cuda(), torch.tensor([1], dtype=torch.int32, device="cuda"),
                                            torch.zeros(3, 3, device="cuda"), True)
    assert int(valid.sum()) == 0
    from prifit_amd.convex_loss import analytic_chamfer_distance
    loss, _ = analytic_chamfer_distance(r, V, c, valid, pts.cuda())
    assert float(loss) == 0.0


def test_speculative_clustering_and_fallback(F):
    """fit_ops.speculative(): same clustering as the synchronous path when round 1 is accepted; a verdict that
    asks for the quantile-doubling retry is reported by spec.ok() and SpeculativeRunner re-runs synchronously."""
    from prifit_amd.train_step import SpeculativeRunner
    _, _, emb = fit_inputs(2, 1024, 32, 5)
    X = emb.cuda()
    ref = F.cluster(X, 0.05, 5, 25)
    with F.speculative() as spec:
        got = F.cluster(X, 0.05, 5, 25)
    assert spec.ok()
    for k in ("bw", "ids", "count", "labels", "Z", "W"):
        assert torch.equal(got[k], ref[k]), k
    # 40 tight prototypes at a small quantile: round 1 finds > 25 clusters
    rng = np.random.default_rng(3)
    proto = rng.normal(size=(40, 32)); proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    e = proto[rng.integers(0, 40, size=(2, 1024))] + 0.003 * rng.normal(size=(2, 1024, 32))
    Xh = _t((e / np.linalg.norm(e, axis=-1, keepdims=True)).astype(np.float32)).cuda()
    with F.speculative() as spec:
        F.cluster(Xh, 0.01, 5, 25)
    assert not spec.ok()

    bn = torch.nn.BatchNorm1d(4).cuda()
    runner = SpeculativeRunner(bn)
    calls = []

    def fn():
        bn(torch.ones(3, 4, device="cuda") * (1.0 + len(calls)))   # a side effect on the running statistics
        calls.append(F._spec is not None)
        return F.cluster(Xh, 0.01, 5, 25)

    before = bn.running_mean.clone()
    out = runner.run(fn, lambda: None)
    want = F.cluster(Xh, 0.01, 5, 25)
    assert calls == [True, False] and runner.fallbacks == 1
    assert torch.equal(out["labels"], want["labels"]) and out["quantile"] == want["quantile"]
    # the discarded attempt left no trace in the buffers: one update, from the second call's input (value 2)
    torch.testing.assert_close(bn.running_mean, before * 0.9 + 0.1 * 2.0)
    assert int(bn.num_batches_tracked) == 1


def test_model_selfsup_step(hiplib):
    """train_partseg_shapenet.py:436-451: forward with include_convex_loss, backward reaches extra_conv_emb and sa1."""
    from prifit_amd.models import pointnet2_part_seg_msg as M
    B, N = 2, 2048
    pts, cham, _ = fit_inputs(B, N, 128, 3)
    torch.manual_seed(1)
    net = M.get_model(50).cuda().train()
    out = net(pts.permute(0, 2, 1).contiguous().cuda(), torch.zeros(B, 1, 16, device="cuda"),
              chamfer_points=cham.permute(0, 2, 1).contiguous().cuda(), include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25)
    assert len(out) == 8 and out[3].shape == (1, 1) and out[4].shape == (1, 1)
    assert abs(net.beta - 0.99) < 1e-12
    loss = out[3].mean()
    assert torch.isfinite(loss)
    loss.backward()
    for name in ("extra_conv_emb.weight", "conv1.weight", "sa1.conv_blocks.0.0.weight", "fp1.mlp_convs.0.weight"):
        gr = dict(net.named_parameters())[name].grad
        assert gr is not None and torch.isfinite(gr).all(), name
    assert len(out[5]) == B and out[5][0].shape == (N,) and out[7].shape == (B, 128, N)


def test_optional_terms_entropy_intersection_pruning(F, golden):
    """include_entropy_loss (pinned vs the reference), include_intersect_loss (parity-unpinned upstream: vs the
    oracle's restatement of the documented intent), include_pruning (a no-op upstream)."""
    from prifit_amd import convex_loss as CLM
    g = golden("fit_entropy")
    seed = int(g["seed"])
    pts, cham, emb = fit_inputs(2, 2048, 128, seed)
    idx = _t(g["idx"])
    # entropy: value vs reference golden; the pre-margin quantity and its gradient vs the oracle
    Xn = torch.nn.functional.normalize(emb, dim=2)
    val = CLM.entropy(Xn[:, idx].cuda())
    torch.testing.assert_close(val.cpu(), _t(g["value"]), rtol=1e-5, atol=1e-6)
    Xo = Xn[:, idx].clone().requires_grad_(True)
    n = idx.shape[0]
    raw_o = torch.stack([((1 + Xo[b] @ Xo[b].t()) ** 2).sum() / n ** 2 for b in range(2)]).mean()
    raw_o.backward()
    Xg = Xn[:, idx].cuda().requires_grad_(True)
    raw = CLM.EntropyFn.apply(Xg)
    raw.backward()
    torch.testing.assert_close(raw.detach().cpu(), raw_o.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(Xg.grad.cpu(), Xo.grad, rtol=1e-4, atol=1e-7)
    # full loss with all optional terms on, vs the oracle with the same explicit random inputs
    jit = _t(synth.uniform01((2, 5000, 3), seed + 11)) * 0.2
    R = _t(synth.uniform01((3, 3), seed))
    kw = dict(quantile=0.05, iterations=10, max_num_clusters=25, include_entropy_loss=True, include_intersect_loss=True,
              alpha=0.01, beta=0.5)
    Xo = emb.permute(0, 2, 1).clone().requires_grad_(True)
    tot_o, ch_o, prm_o, _ = orc.convex_loss(pts.permute(0, 2, 1), cham.permute(0, 2, 1), Xo, rand_table=[[R] * 64] * 2,
                                            canonical=True, entropy_indices=idx, intersect_jitter=jit, **kw)
    tot_o.sum().backward()
    Xg = emb.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    tot, ch, prm, _ = CLM.convex_loss(pts.permute(0, 2, 1).cuda(), cham.permute(0, 2, 1).cuda(), Xg, rand_table=R.cuda(),
                                      canonical=True, entropy_indices=idx.cuda(), intersect_jitter=jit.cuda(),
                                      include_pruning=True, **kw)
    tot.sum().backward()
    assert float(tot_o) > float(ch_o)  # the intersection term is active in this configuration
    torch.testing.assert_close(tot.detach().cpu(), tot_o.detach(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(ch.detach().cpu(), ch_o.detach(), rtol=1e-4, atol=1e-7)
    ref = Xo.grad
    assert abs(Xg.grad.norm().item() - ref.norm().item()) < 1e-2 * ref.norm().item()
    torch.testing.assert_close(Xg.grad.cpu(), ref, rtol=1e-2, atol=1e-3 * ref.abs().max().item())


def test_normalize_twice_kernel_vs_torch(F):
    """convex_loss.py:41,57: the embedding is F.normalize'd twice; one kernel, forward and autograd, incl. a zero row."""
    torch.manual_seed(5)
    x = torch.randn(3, 257, 128)
    x[1, 7] = 0.0
    x[2, 9] *= 1e-14
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.normalize(torch.nn.functional.normalize(xr, dim=2), dim=2)
    g = torch.randn(3, 257, 128)
    (yr * g).sum().backward()
    xd = x.cuda().requires_grad_(True)
    yd = F.Normalize2Fn.apply(xd)
    (yd * g.cuda()).sum().backward()
    torch.testing.assert_close(yd.detach().cpu(), yr.detach(), rtol=1e-6, atol=1e-7)
    ok = x.norm(dim=2) > 1e-10   # rows at the eps clamp: torch differentiates the clamp as constant, same here, but 0/0 noise
    torch.testing.assert_close(xd.grad.cpu()[ok], xr.grad[ok], rtol=1e-4, atol=1e-5)


def test_subsampled_bandwidth_matches_reference_golden(hiplib, golden):
    """num_samples < N bandwidth (src/mean_shift.py:148-151) on the reference's row subset; and the random-subset
    default stays close to it (same statistic, other rows)."""
    from prifit_amd import fit_ops
    g = golden("fit_bandwidth_sub")
    _, _, emb = fit_inputs(2, 2048, 128, int(g["seed"]))
    rows = torch.from_numpy(np.stack([g["rows_%d" % b].astype(np.int64) for b in range(2)]))
    bw = fit_ops.compute_bandwidth(emb.cuda(), 0.05, num_samples=1000, rows=rows.cuda()).cpu()
    ref = torch.tensor([float(g["bw_0"]), float(g["bw_1"])])
    torch.testing.assert_close(bw, ref, rtol=1e-5, atol=1e-7)
    rnd = fit_ops.compute_bandwidth(emb.cuda(), 0.05, num_samples=1000).cpu()
    assert ((rnd - ref).abs() < 0.1 * ref).all()
    with pytest.raises(ValueError):
        fit_ops.compute_bandwidth(emb.cuda(), 0.05, num_samples=1000, rows=rows[:, :10].cuda())


def test_mean_shift_variants_the_loss_never_takes(hiplib, golden):
    """MeanShift.mean_shift_ with the epanechnikov kernel (src/mean_shift.py:70-74) and mean_shift_eff_ (:86-136), both
    unused upstream (eff=False, gaussian), against values captured from the reference: Z and d/dX; and eff=True through
    mean_shift() with the seed rows given."""
    from prifit_amd.src.mean_shift import MeanShift
    g = golden("fit_meanshift_variants")
    seed, N, D = int(g["seed"]), 512, 32
    _, _, emb = fit_inputs(1, N, D, seed, M=1000, noise=0.1)
    X0 = emb[0].cuda()
    G = _t(synth.features(1, N, D, seed + 1))[0].cuda()
    rows = torch.from_numpy(g["rows"].astype(np.int64)).cuda()
    ms, b = MeanShift(), torch.tensor(0.9)
    for name, fn, gg in (("epa", lambda X: ms.mean_shift_(X, b, 3, kernel_type="epa")[0], G),
                         ("eff", lambda X: ms.mean_shift_eff_(X, X[rows], b, 3)[0], G[: N // 2]),
                         ("eff_epa", lambda X: ms.mean_shift_eff_(X, X[rows], b, 3, kernel_type="epa")[0], G[: N // 2])):
        X = X0.clone().requires_grad_(True)
        Z = fn(X)
        (Z * gg).sum().backward()
        torch.testing.assert_close(Z[:64].detach().cpu(), _t(g["Z_" + name]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(Z.detach().sum(0).cpu(), _t(g["Zsum_" + name]), rtol=1e-4, atol=1e-3)
        ref = _t(g["dX_" + name])
        torch.testing.assert_close(X.grad[:64].cpu(), ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item())
        assert abs(X.grad.norm().item() - float(g["dXnorm_" + name])) < 2e-3 * float(g["dXnorm_" + name])
    centre, bw, labels = ms.mean_shift(X0, N, 0.1, 3, eff=True, seed_rows=rows)
    assert centre.shape[1] == D and labels.shape[0] == N // 2 and int(labels.max()) + 1 == centre.shape[0]


def test_nms_with_distinct_centres_and_epanechnikov_guard(hiplib, golden):
    """MeanShift.nms(centers, X, b) with centres that are not the points (src/mean_shift.py:162-202; shifted points against
    the original embedding) and guard_mean_shift / kernel_type="epa" (src/ellipsoid_utils.py:9-27 -> src/mean_shift.py:70-74)
    against the reference's ids / labels / retry sequence (tests/golden/fit_nms_pair.npz), and against the oracle on
    2048 x 128 inputs."""
    from prifit_amd.src.mean_shift import MeanShift
    from prifit_amd.src import ellipsoid_utils as EU
    g = golden("fit_nms_pair")
    seed, N, D = int(g["seed"]), 512, 32
    _, _, emb = fit_inputs(2, N, D, seed, M=1000, noise=0.1)
    ms = MeanShift()
    for b in range(2):
        X = emb[b]
        bw = _t(g["bw_%d" % b])
        Z = orc.mean_shift_iterations(X, bw, 4)          # the same centres on both sides: this test is about nms
        kept, ids, labels = ms.nms(Z.cuda(), X.cuda(), bw.cuda())
        assert torch.equal(ids.cpu(), _t(g["ids_%d" % b]).long()) and torch.equal(labels.cpu(), _t(g["labels_%d" % b]).long())
        torch.testing.assert_close(kept.cpu(), Z[ids.cpu()], rtol=0, atol=0)
    # a larger case against the oracle (128-d, 2048 points: the chord kernels' tiled shapes)
    _, _, big = fit_inputs(1, 2048, 128, 12, noise=0.05)
    X = big[0]
    bw = orc.compute_bandwidth(X, 0.05)
    Z = orc.mean_shift_iterations(X, bw, 3)
    _, io, lo = orc.nms(Z, X, bw)
    _, ids, labels = ms.nms(Z.cuda(), X.cuda(), bw)
    assert torch.equal(ids.cpu(), io) and torch.equal(labels.cpu(), lo)
    with pytest.raises(RuntimeError):
        ms.nms(Z[:100].cuda(), X.cuda(), bw)              # upstream's broadcast (:191) fails the same way
    # epanechnikov kernel through the guard: the reference's retry sequence ends at the same quantile / K / labels
    centre, bw, labels = EU.guard_mean_shift(emb[0].cuda(), N, float(g["q0"]), int(g["iters"]), int(g["cap"]), kernel_type="epa")
    assert centre.shape[0] == int(g["epa_K"])
    assert abs(float(bw) - float(g["epa_bw"])) <= 1e-5 * float(g["epa_bw"])
    # (K = 5 after four doublings: a broad kernel over modes that are not well separated.  The labels are nearest-
    # REPRESENTATIVE assignments, and which point represents a mode is rounding noise in upstream's nms: another
    # representative moves the boundary between two modes -- measured 3 % of the points.  K, the bandwidth and the retry sequence
    # are the reference's; the partition up to those boundary points.)
    from dgcnn_common import partition_agreement
    assert partition_agreement(labels.cpu(), _t(g["epa_labels"]).long()) >= 0.9


def test_nms_owner_pass_fused_into_the_chord_kernel(F, monkeypatch):
    """nms with the owner pass (argmin over every column of 2 - 2 Z Z^T, first minimum) taken in the chord kernel's epilogue
    through 64-bit atomic-min keys against the separate pass that re-reads the matrix: the very same owners, kept ids,
    labels -- on clustered points, on a collapsed cloud (all points within 1e-7: ties and negative distances everywhere)
    and with exact duplicate rows in different tiles."""
    gen = torch.Generator().manual_seed(9)
    _, _, emb = fit_inputs(3, 2048, 128, 4)
    Zs = [emb]
    base = torch.nn.functional.normalize(torch.randn(1, 1, 128, generator=gen), dim=2)
    Zs.append(torch.nn.functional.normalize(base + 1e-7 * torch.randn(2, 1024, 128, generator=gen), dim=2))
    dup = torch.nn.functional.normalize(torch.randn(2, 512, 64, generator=gen), dim=2)
    dup[:, 300:340] = dup[:, 7:47]            # duplicates 293 rows apart: other tiles
    dup[:, 129] = dup[:, 0]
    Zs.append(dup)
    for Z in Zs:
        Zc = Z.cuda().contiguous()
        bw = torch.full((Z.shape[0],), 0.3, device="cuda")
        res = []
        # round 6: with the owner pass fused in, nms reads one bit per element -- the default forms a bit mask (dist < b) in the
        # chord kernel and never writes the matrix (NMS_MASK); per-shape bandwidths, one so small that no neighbour qualifies
        for bws in (bw, torch.tensor([0.3, 1e-9, 0.7][:Z.shape[0]], device="cuda")):
            res = []
            for fused, mask in ((True, True), (True, False), (False, False)):
                monkeypatch.setattr(F, "FUSE_NMS_OWNER", fused)
                monkeypatch.setattr(F, "NMS_MASK", mask)
                res.append(F.nms(Zc, bws))
            for other in res[1:]:
                for a, b in zip(res[0], other):
                    assert torch.equal(a, b)


def test_bandwidth_with_more_samples_than_rows(hiplib, golden):
    """num_samples > N keeps all rows and takes K = int(quantile * num_samples) (src/mean_shift.py:151-155), against the
    value captured from the reference; K beyond the row length raises like upstream's topk."""
    from prifit_amd import fit_ops
    from prifit_amd.src.mean_shift import MeanShift
    g = golden("fit_bandwidth_over")
    _, _, emb = fit_inputs(2, int(g["N"]), 128, int(g["seed"]))
    bw = fit_ops.compute_bandwidth(emb.cuda(), float(g["quantile"]), num_samples=int(g["num_samples"])).cpu()
    ref = torch.tensor([float(g["bw_0"]), float(g["bw_1"])])
    torch.testing.assert_close(bw, ref, rtol=1e-5, atol=1e-7)
    one = MeanShift().compute_bandwidth(emb[0].cuda(), int(g["num_samples"]), float(g["quantile"]))
    torch.testing.assert_close(one.cpu(), ref[0], rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        fit_ops.compute_bandwidth(emb.cuda(), 0.9, num_samples=1000)      # K = 900 > 512 rows


@pytest.mark.parametrize("N,D,T,nrows", [(2048, 128, 10, (8, 1, 32)), (300, 128, 3, (5, 0, 2)), (1500, 32, 5, (3, 32, 7)),
                                         (256, 64, 2, (32, 32, 32)), (2048, 128, 0, (4, 4, 4)),
                                         (2048, 128, 10, (25, 40, 64)), (700, 64, 4, (64, 33, 0))])
def test_mean_shift_row_sparse_backward_equals_dense(F, N, D, T, nrows):
    """MeanShiftRowsFn (the loss reads new_X[ids] only: backward on those rows alone) against the dense engine --
    MeanShiftFn + gather, whose other rows multiply exact zeros: same centres bit for bit, same dX to fp32 rounding.
    Ragged N, all three widths, per-shape live counts incl. 0 and the full 32 slots, zero iterations; round 6: 64 slots per
    shape (max_num_clusters above 32: the kernels' RMAX = 64 instantiation), 25 / 40 / 64 live rows.  Every case runs twice:
    the last-ticket hand-over between workgroups must give the same bits from run to run."""
    B, R = 3, (F.KM if max(nrows) <= F.KM else F.KM_MAX)
    gen = torch.Generator().manual_seed(N + D + T)
    # clustered rows so that kernel values span the clamp: prototypes + noise
    proto = torch.nn.functional.normalize(torch.randn(6, D, generator=gen), dim=1)
    X = torch.nn.functional.normalize(proto[torch.randint(0, 6, (B, N), generator=gen)] + 0.15 * torch.randn(B, N, D, generator=gen), dim=2)
    bw = torch.tensor([0.35, 0.5, 0.8])
    ids = torch.stack([torch.randperm(N, generator=gen)[:R] for _ in range(B)])
    nr = torch.tensor(nrows, dtype=torch.int32)
    G = torch.randn(B, R, D, generator=gen)
    live = (torch.arange(R).view(1, R) < nr.view(B, 1)).view(B, R, 1).float()

    Xd = X.cuda().requires_grad_(True)
    Zd = F.MeanShiftFn.apply(Xd, bw.cuda(), T) if T else Xd.clone()
    cd = torch.gather(Zd, 1, ids.cuda().unsqueeze(-1).expand(-1, -1, D))
    (cd * (G * live).cuda()).sum().backward()

    Xr = X.cuda().requires_grad_(True)
    with torch.no_grad():
        Zf, traj = F.mean_shift_trajectory(Xr.detach(), bw.cuda(), T, keep_kernel=False)
    assert all(it[1] is None for it in traj)
    cr = F.MeanShiftRowsFn.apply(Xr, bw.cuda(), ids.cuda(), nr.cuda(), traj)
    (cr * G.cuda()).sum().backward()          # slots beyond nrows: their gradient must be ignored

    assert torch.equal(Zf, Zd.detach())
    assert torch.equal(cr.detach(), cd.detach())
    ref = Xd.grad
    assert torch.isfinite(Xr.grad).all()
    torch.testing.assert_close(Xr.grad, ref, rtol=2e-4, atol=2e-5 * ref.abs().max().item())
    assert abs(Xr.grad.norm().item() - ref.norm().item()) <= 1e-4 * ref.norm().item()
    # fixed summation orders (the workgroup that happens to finish last in a shape sums the tile slabs in tile order): same bits
    X2 = X.cuda().requires_grad_(True)
    with torch.no_grad():
        _, traj2 = F.mean_shift_trajectory(X2.detach(), bw.cuda(), T, keep_kernel=False)
    (F.MeanShiftRowsFn.apply(X2, bw.cuda(), ids.cuda(), nr.cuda(), traj2) * G.cuda()).sum().backward()
    assert torch.equal(X2.grad, Xr.grad)
    # round 6: up to 32 slots the default (mode 2) gives every live row a workgroup of its own that runs all T iterations (no
    # hand-over between workgroups at all); mode 1 (the default above 32 slots) runs the key-tiled iterations as one launch
    # over a work queue; mode 0 = one launch per iteration.  Modes 0 and 1 compute the same things in the same order: same
    # bits.  Mode 2 sums over the keys in another order: fp32 rounding.
    assert F.MS_ROWS_MODE == "auto" and F.rows_mode(32) == 2 and F.rows_mode(64) == 1
    got = {}
    for mode in (0, 1, 2):
        Xm = X.cuda().requires_grad_(True)
        F.MS_ROWS_MODE = str(mode)
        try:
            (F.MeanShiftRowsFn.apply(Xm, bw.cuda(), ids.cuda(), nr.cuda(), traj2_copy(F, Xm, bw, T)) * G.cuda()).sum().backward()
        finally:
            F.MS_ROWS_MODE = "auto"
        got[mode] = Xm.grad
    assert torch.equal(got[0], got[1])
    assert torch.equal(got[F.rows_mode(R)], Xr.grad)
    torch.testing.assert_close(got[2], got[0], rtol=1e-4, atol=2e-6 * ref.abs().max().item())


def traj2_copy(F, X, bw, T):
    with torch.no_grad():
        return F.mean_shift_trajectory(X.detach(), bw.cuda(), T, keep_kernel=False)[1]


def test_cluster_gradient_same_with_both_mean_shift_engines(F, monkeypatch):
    """cluster() -> centres / membership -> a scalar: d/dX with the row-sparse engine (default) and with the dense one.
    (Both on the standard forward kernel: the row-sparse engine's first update otherwise reads the chord matrix, whose s differs
    from the product's by one rounding -- test_first_update_from_the_chord_matrix -- and the forward is compared bit for bit.)"""
    monkeypatch.setattr(F, "MS_FIRST_CHORD", False)
    _, _, emb = fit_inputs(2, 2048, 128, 7)
    Gc = torch.randn(2, F.KM, 128, generator=torch.Generator().manual_seed(1)).cuda()
    Gw = torch.randn(2, 2048, F.KM, generator=torch.Generator().manual_seed(2)).cuda()
    grads, outs = [], []
    for rows in (True, False):
        monkeypatch.setattr(F, "ROWS_BWD", rows)
        X = emb.cuda().requires_grad_(True)
        cl = F.cluster(X, 0.05, 10, 25)
        live = (torch.arange(F.KM, device="cuda").view(1, -1) < cl["count"].view(-1, 1)).unsqueeze(-1)
        ((cl["centres"] * Gc * live).sum() + (cl["W"] * Gw).sum()).backward()
        grads.append(X.grad.clone())
        outs.append(cl)
    for k in ("bw", "ids", "count", "labels", "Z", "centres", "W"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert int(outs[0]["count"].min()) >= 2
    torch.testing.assert_close(grads[0], grads[1], rtol=2e-4, atol=2e-5 * grads[1].abs().max().item())


def test_dx_streams_kernel_matches_dual_gemm(hiplib):
    """prifit_meanshift_dx_streams (opt-in, atomics-free dX of a mean-shift backward iteration) against the dual-source
    GEMM on the same streams, and its run-to-run bit-reproducibility."""
    import ctypes
    from prifit_amd._lib import call, ptr, cur_stream
    LL = ctypes.c_longlong
    B, N, D = 3, 256, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    gS = torch.randn(B, N, N, device="cuda", generator=g)
    Kt = torch.rand(B, N, N, device="cuda", generator=g)
    Z = torch.randn(B, N, D, device="cuda", generator=g)
    gO = torch.randn(B, N, D, device="cuda", generator=g)
    ref = (gS.double() @ Z.double() + Kt.double() @ gO.double()).float()      # rows = keys: gS^T / K^T are stored [key][query]
    outs = []
    for _ in range(2):
        dX = torch.zeros(B, N, D, device="cuda")
        call("prifit_meanshift_dx_streams", ptr(gO), ptr(Z), ptr(gS), ptr(Kt), LL(N), LL(N * N), B, N, D, ptr(dX), cur_stream())
        outs.append(dX)
    assert torch.equal(outs[0], outs[1])
    tol = 2e-5 * (gS.norm(dim=2, keepdim=True) * Z.norm(dim=1).unsqueeze(1).amax(dim=2, keepdim=True) +
                  Kt.norm(dim=2, keepdim=True) * gO.norm(dim=1).unsqueeze(1).amax(dim=2, keepdim=True))
    assert ((outs[0] - ref).abs() <= tol + 1e-5).all()
    d2 = torch.zeros(B, N, D, device="cuda")
    call("prifit_gemm_dual_nn_f32", N, D, N, N, ptr(gS), ptr(Kt), LL(N), LL(N * N), ptr(Z), ptr(gO), LL(D), LL(N * D), ptr(d2),
         LL(D), LL(N * D), B, 1, 1, cur_stream())
    torch.testing.assert_close(outs[0], d2, rtol=1e-4, atol=1e-3)


def test_stream_k_schedules_of_the_fused_mean_shift_kernels(hiplib):
    """The stream-K (balanced) form of the dense backward's dZ kernel against its plain-grid form on a size where the plain
    grid leaves a partial round (B x N/64 > resident slots, not a multiple of them): same dZ to rounding (split query
    blocks add two partial sums), identical gS^T stream.  (The forward has no such schedule.)"""
    import ctypes
    from prifit_amd._lib import call, ptr, cur_stream
    LL = ctypes.c_longlong
    B, N, D = 35, 1024, 128            # 560 query blocks for 512 slots
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.nn.functional.normalize(torch.randn(B, N, D, device="cuda", generator=g), dim=2)
    Z = torch.nn.functional.normalize(X + 0.1 * torch.randn(B, N, D, device="cuda", generator=g), dim=2)
    bw = torch.full((B,), 0.5, device="cuda")
    res = []
    for bal in (0, 1):
        KT = torch.empty(B, N, N, device="cuda")
        Zn, O = torch.empty_like(Z), torch.zeros_like(Z)
        rs, nrm = torch.zeros(B, N, device="cuda"), torch.empty(B, N, device="cuda")
        call("prifit_meanshift_fused_fwd", ptr(Z), ptr(X), ptr(bw), B, N, D, ptr(KT), LL(N), LL(N * N), ptr(Zn), ptr(O), ptr(rs),
             ptr(nrm), cur_stream())
        gO = torch.randn(B, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
        grs = torch.randn(B, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        gS, dZ = torch.empty(B, N, N, device="cuda"), torch.zeros(B, N, D, device="cuda")
        call("prifit_meanshift_fused_bwd_dz", ptr(gO), LL(N * D), ptr(X), ptr(bw), ptr(grs), ptr(KT), LL(N), LL(N * N), ptr(gS),
             B, N, D, ptr(dZ), bal, cur_stream())
        res.append((KT, Zn, O, rs, nrm, gS, dZ))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[5], b[5])                 # the N x N streams: element for element
    for i, tol in ((1, 1e-6), (2, 1e-4), (3, 1e-4), (4, 1e-6), (6, 1e-4)):
        torch.testing.assert_close(a[i], b[i], rtol=1e-5, atol=tol)


def test_split_mean_shift_products_error_against_fp64(F, golden, monkeypatch):
    """The labelled experiment (fit_ops.MS_SPLIT): ten mean-shift updates on the golden's embedding with the forward
    products as fp32 MFMA (the product path), bf16x3, bf16x6 and fp16x3, each against the same updates in fp64.  The
    fp32-grade forms must stay within a small factor of the fp32 kernel's own error and within the golden's tolerances;
    bf16x3 is reported only (its operands carry 16 bits)."""
    g = golden("fit_meanshift")
    _, _, emb = fit_inputs(2, 2048, 128, int(g["seed"]))
    bw = torch.stack([_t(g["bw_0"]), _t(g["bw_1"])]).cuda()
    X = emb.cuda().contiguous()
    X64, Z64 = X.double(), X.double()
    for _ in range(10):   # src/mean_shift.py:61-82 in fp64
        S = Z64 @ X64.transpose(1, 2)
        K = torch.exp(torch.clamp((S - 1.0) / (bw.double() ** 2)[:, None, None], -13.0, 75.0))
        new = Z64 + (K @ X64 / K.sum(-1, keepdim=True) - Z64)
        Z64 = new / new.norm(dim=-1, keepdim=True)
    err = {}
    for mode in ("0", "bf16x3", "bf16x6", "fp16x3"):
        monkeypatch.setattr(F, "MS_SPLIT", mode)
        Z, traj = F.mean_shift_trajectory(X, bw, 10, False)
        assert len(traj) == 10
        err[mode] = ((Z.double() - Z64).abs().max() / Z64.abs().max()).item()
        if mode != "bf16x3":
            for b in range(2):
                torch.testing.assert_close(Z[b, :64].cpu(), _t(g[f"Z_head_{b}"]), rtol=1e-4, atol=1e-5)
                torch.testing.assert_close(Z[b].sum(0).cpu(), _t(g[f"Z_colsum_{b}"]), rtol=1e-4, atol=1e-3)
    print("max |Z - Z_fp64| / max |Z_fp64| after 10 updates:", {k: "%.2e" % v for k, v in err.items()})
    assert err["bf16x6"] < 4 * err["0"] + 1e-7 and err["fp16x3"] < 8 * err["0"] + 1e-7
    assert err["bf16x3"] < 1e-2


def test_split_mode_falls_back_where_it_does_not_apply(F, monkeypatch):
    monkeypatch.setattr(F, "MS_SPLIT", "bf16x6")
    assert F.split_mode(2048, 128) == 2 and F.split_mode(2048, 128, keep_kernel=True) == 0
    assert F.split_mode(300, 128) == 0 and F.split_mode(2048, 64) == 0
    monkeypatch.setattr(F, "MS_SPLIT", "bf16x9")
    with pytest.raises(KeyError):
        F.split_mode(2048, 128)


@pytest.mark.parametrize("B,N", [(8, 256), (3, 512), (16, 1024)])
def test_split_products_one_update_all_mappings(F, B, N):
    """One update of the labelled experiment's kernels against fp64 on small shapes: B % 8 == 0 takes the XCD-aware block
    mapping, other batch sizes the plain one; N = 256 is a single query block per shape."""
    from prifit_amd._lib import call, cur_stream, dll, ptr
    X = torch.nn.functional.normalize(_t(synth.features(B, N, 128, 40 + B)), dim=-1).cuda()
    Z = torch.nn.functional.normalize(X + 0.05 * _t(synth.features(B, N, 128, 41 + B)).cuda(), dim=-1).contiguous()
    bw = torch.linspace(0.6, 1.1, B).cuda()
    S = Z.double() @ X.double().transpose(1, 2)
    K = torch.exp(torch.clamp((S - 1.0) / (bw.double() ** 2)[:, None, None], -13.0, 75.0))
    O64, r64 = K @ X.double(), K.sum(-1)
    for mode, tol in ((1, 3e-5), (2, 3e-6), (3, 3e-6)):
        assert dll().prifit_meanshift_split_supported(N, 128, mode) == 1
        cut = torch.empty(dll().prifit_meanshift_split_workspace(B, N, 128, mode), dtype=torch.uint8, device="cuda")
        call("prifit_meanshift_split_prep", ptr(X), B, N, 128, mode, ptr(cut), cur_stream())
        O = torch.full((B, N, 128), float("nan"), device="cuda")
        rs = torch.full((B, N), float("nan"), device="cuda")
        call("prifit_meanshift_split_fwd", ptr(Z), ptr(cut), ptr(bw), B, N, 128, mode, ptr(O), ptr(rs), cur_stream())
        assert ((O.double() - O64).abs().amax(dim=(1, 2)) / O64.abs().amax(dim=(1, 2))).max().item() < tol
        assert ((rs.double() - r64).abs() / r64).max().item() < tol
    assert dll().prifit_meanshift_split_supported(300, 128, 2) == 0 and dll().prifit_meanshift_split_supported(256, 64, 2) == 0
    assert dll().prifit_meanshift_split_supported(256, 128, 7) == 0


def test_first_update_from_the_chord_matrix(F):
    """The first mean-shift update of a trajectory reads S = X X^T from the chord matrix the bandwidth step wrote
    (prifit_meanshift_fused_first_fwd, PRIFIT_MS_FIRST_CHORD) instead of forming it again: every saved tensor of that update
    (O, row sums, norms, the new points) and the end point of ten updates against the standard kernel.  Measured bar: the two
    forms of s differ by one rounding of (1 - d), i.e. ~6e-8 absolute in s and up to 6e-8 / b^2 relative in a kernel value."""
    B, N, D = 3, 2048, 128
    gen = torch.Generator().manual_seed(5)
    proto = torch.nn.functional.normalize(torch.randn(8, D, generator=gen), dim=1)
    X = torch.nn.functional.normalize(proto[torch.randint(0, 8, (B, N), generator=gen)] + 0.05 * torch.randn(B, N, D, generator=gen), dim=2).cuda()
    keep = []
    bw = F.compute_bandwidth(X, 0.05, keep_chord=keep)
    assert len(keep) == 1 and keep[0].shape == (B, N, N)
    assert F.compute_bandwidth(X, 0.05, num_samples=512, rows=torch.arange(512).repeat(B, 1), keep_chord=keep) is not None and len(keep) == 1
    Za, sa = F.mean_shift_trajectory(X, bw, 10, keep_kernel=False, chord=keep[0])
    Zb, sb = F.mean_shift_trajectory(X, bw, 10, keep_kernel=False)
    for name, i, tol in (("O", 2, 2e-5), ("rowsum", 3, 2e-5), ("Z1", 4, 2e-6), ("nrm", 5, 2e-5)):
        a, b = sa[0][i], sb[0][i]
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))
    assert float((Za - Zb).abs().max()) <= 5e-6
    # the dense engine (K^T kept) and the switch take the standard kernel
    old = F.MS_FIRST_CHORD
    F.MS_FIRST_CHORD = False
    try:
        Zc, _ = F.mean_shift_trajectory(X, bw, 10, keep_kernel=False, chord=keep[0])
    finally:
        F.MS_FIRST_CHORD = old
    assert torch.equal(Zc, Zb)
