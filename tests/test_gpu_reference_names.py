"""GPU: the reference's OWN names and signatures, called the way upstream code calls them, against the oracle and the
goldens captured from the reference (VERDICT r3 "missing" 2-3, "weak" 2-3):

* the free functions of models/pointnet_util.py:19-157 -- square_distance, index_points (2-D and 3-D idx),
  farthest_point_sample(xyz, npoint) with its random start, query_ball_point(radius, nsample, xyz, new_xyz),
  sample_and_group(..., returnfps=True), sample_and_group_all;
* convex_loss.py:313-343, :374-413, :444-502 -- compute_sdf_ellipsoid(s)(_batch), compute_sdf_cuboid(s)(_batch),
  compute_intersection_loss_volume_3, prune_points;
* convex_loss.py:106, :163, :227, :285, :346, :416 -- the intersection variants upstream keeps but never calls, and
  sample_axis, value and gradient against the reference's autograd;
* src/ellipsoid_utils.py:162-214 sample_from_pred_params_cuboid, src/sample_ellipsoid.py:65-96 sample_cuboid,
  src/ellipsoid_fitting.py:19-69,119-141 single-cluster weighted_ellipsoid_fitting / principal_axis_ellipsoid;
* the gradient through `center = new_X[indices]` alone (src/mean_shift.py:46) on the row-sparse engine against the
  reference's autograd.
Everything is imported through compat.install(), i.e. by the reference's module paths."""
import importlib

import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth
from tests_helpers import check_intersection_grads, fit_inputs, intersection_case

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ref_names(hiplib):
    assert torch.cuda.is_available()
    from prifit_amd import compat
    compat.install()
    return importlib.import_module


def test_pointnet_util_free_functions_by_reference_signature(ref_names):
    PU = ref_names("models.pointnet_util")
    B, N, S, K, D = 3, 1024, 128, 32, 5
    xyz = _t(synth.cloud("surface", B, N, 21))
    feat = _t(synth.features(B, N, D, 22))
    xd, fd = xyz.cuda(), feat.cuda()
    # farthest_point_sample(xyz, npoint): the start index is random upstream (:75); whatever it drew, the chain that
    # follows must be the oracle's chain from that start -- and a given start reproduces bit for bit
    fps = PU.farthest_point_sample(xd, S)
    assert fps.dtype == torch.int64 and fps.shape == (B, S)
    assert torch.equal(fps.cpu(), orc.c_farthest_point_sample(xyz, S, fps[:, 0].cpu()))
    # index_points with 2-D and 3-D indices (:43-60)
    new_xyz = PU.index_points(xd, fps)
    assert torch.equal(new_xyz.cpu(), orc.gather_rows(xyz, fps.cpu()))
    # query_ball_point(radius, nsample, xyz, new_xyz) (:87-107): int64, bit-exact
    idx = PU.query_ball_point(0.3, K, xd, new_xyz)
    assert idx.dtype == torch.int64
    want_idx = orc.c_query_ball_point(0.3, K, xyz, new_xyz.cpu())
    assert torch.equal(idx.cpu(), want_idx)
    g3 = PU.index_points(fd, idx)
    assert g3.shape == (B, S, K, D)
    assert torch.equal(g3.cpu(), orc.gather_rows(feat, want_idx.reshape(B, -1)).reshape(B, S, K, D))
    # square_distance (:19-40): expanded form, bitwise
    assert torch.equal(PU.square_distance(new_xyz, xd).cpu(), orc.c_square_distance(new_xyz.cpu(), xyz))
    # sample_and_group(..., returnfps=True) (:110-137): [rel_xyz | features], grouped_xyz, fps_idx
    torch.manual_seed(5)
    nx, npts, gxyz, fidx = PU.sample_and_group(S, 0.3, K, xd, fd, returnfps=True)
    start = fidx[:, 0].cpu()
    want_f = orc.c_farthest_point_sample(xyz, S, start)
    assert torch.equal(fidx.cpu(), want_f)
    want_c = orc.gather_rows(xyz, want_f)
    assert torch.equal(nx.cpu(), want_c)
    wi = orc.c_query_ball_point(0.3, K, xyz, want_c)
    want_g = orc.gather_rows(xyz, wi.reshape(B, -1)).reshape(B, S, K, 3)
    assert torch.equal(gxyz.cpu(), want_g)
    want_new = torch.cat([want_g - want_c.view(B, S, 1, 3), orc.gather_rows(feat, wi.reshape(B, -1)).reshape(B, S, K, D)], -1)
    assert npts.shape == (B, S, K, 3 + D)
    assert torch.equal(npts.cpu(), want_new)
    nx2, np2 = PU.sample_and_group(S, 0.3, K, xd, None)          # points=None: rel_xyz only
    assert np2.shape == (B, S, K, 3)
    # sample_and_group_all (:140-157)
    z, allp = PU.sample_and_group_all(xd, fd)
    assert torch.equal(z.cpu(), torch.zeros(B, 1, 3)) and allp.shape == (B, 1, N, 3 + D)
    assert torch.equal(allp.cpu(), torch.cat([xyz.view(B, 1, N, 3), feat.view(B, 1, N, D)], -1))
    assert torch.equal(PU.sample_and_group_all(xd, None)[1].cpu(), xyz.view(B, 1, N, 3))


def _golden_params(g, dev=None):
    out = []
    for b in range(3):
        out.append([tuple(_t(g[f"{n}_{b}"][k]).to(dev) if dev else _t(g[f"{n}_{b}"][k]) for n in ("r", "V", "c"))
                    for k in range(g[f"r_{b}"].shape[0])])
    return out


def test_prune_points_matches_reference_golden(ref_names, golden):
    CL = ref_names("convex_loss")
    EU = ref_names("src.ellipsoid_utils")
    g = golden("fit_prune")
    params = _golden_params(g, "cuda")
    params_cpu = _golden_params(g)
    pts = EU.sample_from_pred_params(params, 500)
    want_pts = orc.sample_from_params(params_cpu)
    kept = CL.prune_points(pts, params)
    for b in range(3):
        assert pts[b].shape[0] == int(g[f"n_{b}"])
        torch.testing.assert_close(pts[b].cpu(), want_pts[b], rtol=1e-5, atol=2e-6)
        keep_ref, m = _t(g[f"keep_{b}"]), _t(g[f"minsdf_{b}"])
        # the mask is a threshold on an fp32 SDF: a point within rounding distance of the threshold may fall either way
        sure = (m + 1e-3).abs() > 1e-5
        with torch.no_grad():
            mine = torch.stack(CL.compute_sdf_ellipsoids(pts[b], params[b]), 1).min(1)[0].cpu()
        torch.testing.assert_close(mine, m, rtol=1e-4, atol=1e-5)
        keep_mine = mine > -1e-3
        assert torch.equal(keep_mine[sure], keep_ref[sure])
        assert kept[b].shape[0] == int(keep_mine.sum()) and abs(kept[b].shape[0] - int(keep_ref.sum())) <= int((~sure).sum())
        assert torch.equal(kept[b].cpu(), pts[b].cpu()[keep_mine])
    # gradients flow through the kept points to the parameters (upstream: only the mask is under no_grad)
    r = params[0][0][0].clone().requires_grad_(True)
    prm = [[(r,) + params[0][0][1:]] + params[0][1:]]
    p1 = EU.sample_from_pred_params(prm, 500)
    CL.prune_points(p1, prm)[0].sum().backward()
    assert r.grad is not None and torch.isfinite(r.grad).all() and r.grad.abs().sum() > 0


@pytest.mark.parametrize("cuboid", [False, True])
def test_sdf_and_intersection_by_reference_names(ref_names, golden, cuboid):
    CL = ref_names("convex_loss")
    g = golden("fit_prune")
    params, params_cpu = _golden_params(g, "cuda"), _golden_params(g)
    pts = _t(synth.cloud("cube", 3, 700, 9)) * 0.6
    pd = pts.cuda()
    one = (CL.compute_sdf_cuboid if cuboid else CL.compute_sdf_ellipsoid)
    many = (CL.compute_sdf_cuboids if cuboid else CL.compute_sdf_ellipsoids)
    batch = (CL.compute_sdf_cuboid_batch if cuboid else CL.compute_sdf_ellipsoids_batch)
    fn = orc.sdf_cuboid if cuboid else orc.sdf_ellipsoid
    r, V, c = params[1][2]
    rc, Vc, cc = params_cpu[1][2]
    torch.testing.assert_close(one(pd[1], c, r, V).cpu(), fn(pts[1], cc, rc, Vc), rtol=1e-4, atol=1e-5)
    got = many(pd[0], params[0])
    assert len(got) == len(params[0])
    for k, (rk, Vk, ck) in enumerate(params_cpu[0]):
        torch.testing.assert_close(got[k].cpu(), fn(pts[0], ck, rk, Vk), rtol=1e-4, atol=1e-5)
    gb = batch(pd, params)
    assert [len(x) for x in gb] == [3, 5, 1]
    torch.testing.assert_close(gb[1][4].cpu(), fn(pts[1], params_cpu[1][4][2], params_cpu[1][4][0], params_cpu[1][4][1]),
                               rtol=1e-4, atol=1e-5)
    # compute_intersection_loss_volume_3(ellipsoid_params_batch, points, cuboid) (:374-413; upstream broken, G7: vs the oracle)
    want = orc.intersection_loss_volume_3(params_cpu, pts, cuboid=cuboid)
    got = CL.compute_intersection_loss_volume_3(params, pd, cuboid=cuboid)
    torch.testing.assert_close(got.cpu().reshape(()), want.reshape(()), rtol=1e-4, atol=1e-7)


def test_cuboid_sampler_by_reference_names(ref_names, golden):
    EU = ref_names("src.ellipsoid_utils")
    SE = ref_names("src.sample_ellipsoid")
    g = golden("fit_prune")
    params, params_cpu = _golden_params(g, "cuda"), _golden_params(g)
    got = EU.sample_from_pred_params_cuboid(params + [[]], 500)
    want = orc.sample_from_params(params_cpu + [[]], cuboid=True)
    assert got[3] == -1 and want[3] == -1
    for b in range(3):
        assert got[b].shape == want[b].shape
        torch.testing.assert_close(got[b].cpu(), want[b], rtol=1e-5, atol=2e-6)
    r, V, c = params[0][1]
    rc, Vc, cc = params_cpu[0][1]
    a = r.clone().requires_grad_(True)
    p, _ = SE.SampleEllipsoid().sample_cuboid(a[0], a[1], a[2], c, V, n=777)
    torch.testing.assert_close(p.detach().cpu(), orc.sample_cuboid(rc[0], rc[1], rc[2], cc, Vc, 777), rtol=1e-5, atol=2e-6)
    p.sum().backward()
    assert torch.isfinite(a.grad).all() and a.grad.abs().sum() > 0


def test_single_cluster_fit_by_reference_names(ref_names):
    EF = ref_names("src.ellipsoid_fitting")
    pts, _, _ = fit_inputs(1, 2048, 32, 4)
    P = pts[0]
    w = torch.exp(-((P - P[:300].mean(0)) ** 2).sum(1, keepdim=True) / 0.05) + 1e-3      # one soft cluster
    R = _t(synth.uniform01((3, 3), 3))
    r0, V0, c0 = orc.fit_ellipsoid(P, w, R, canonical=True)
    r, V, c = EF.weighted_ellipsoid_fitting(P.cuda(), w.cuda(), rand_table=R.cuda(), canonical=True)
    torch.testing.assert_close(r.cpu(), r0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(c.cpu(), c0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(V.cpu(), V0, rtol=1e-3, atol=1e-3)
    # principal_axis_ellipsoid(points, weights, S, V, mode) (:119-141) with a caller-supplied SVD
    Pc = P - (P * w).sum(0) / w.sum()
    S = torch.linalg.svdvals((Pc * w).T @ Pc / w.sum())
    ax, Vout = EF.principal_axis_ellipsoid(Pc.cuda(), w.cuda(), S.cuda(), V0.cuda(), mode="slow")
    torch.testing.assert_close(ax.cpu(), r0, rtol=1e-4, atol=1e-5)
    fast, _ = EF.principal_axis_ellipsoid(Pc.cuda(), w.cuda(), S.cuda(), V0.cuda(), mode="fast")
    torch.testing.assert_close(fast.cpu(), torch.sqrt(S.clamp(min=1e-7)) * 1.732)
    # a degenerate cluster (all weight on one line) is rejected like upstream (:43-47): -1
    line = torch.zeros(2048, 3)
    line[:, 0] = torch.linspace(-1, 1, 2048)
    assert EF.weighted_ellipsoid_fitting(line.cuda(), torch.ones(2048, 1).cuda(), rand_table=torch.zeros(3, 3).cuda()) == -1


def test_center_gather_gradient_matches_reference_golden(ref_names, golden):
    """The row-sparse mean-shift backward (fit_ops.MeanShiftRowsFn) pinned DIRECTLY against the reference's autograd
    through `center = new_X[indices]` (src/mean_shift.py:44-46): 2e-3 of the largest entry, norm 1e-3."""
    from prifit_amd import fit_ops as F
    g = golden("fit_center_grad")
    seed, T = int(g["seed"]), int(g["iterations"])
    _, _, emb = fit_inputs(2, 2048, 128, seed)
    R = max(g["ids_0"].shape[0], g["ids_1"].shape[0])
    ids = torch.zeros(2, F.KM, dtype=torch.int64)
    Gc = torch.zeros(2, F.KM, 128)
    nrows = torch.zeros(2, dtype=torch.int32)
    for b in range(2):
        k = g[f"ids_{b}"].shape[0]
        ids[b, :k] = _t(g[f"ids_{b}"]).long()
        Gc[b, :k] = _t(g[f"G_{b}"])
        nrows[b] = k
    assert R <= F.KM
    X = emb.cuda().contiguous().requires_grad_(True)
    bw = torch.stack([_t(g["bw_0"]), _t(g["bw_1"])]).float().cuda()
    assert F.ROWS_BWD and F.rows_supported(2048, 128, F.KM)
    with torch.no_grad():
        _, traj = F.mean_shift_trajectory(X.detach(), bw, T, keep_kernel=False)
    cen = F.MeanShiftRowsFn.apply(X, bw, ids.cuda(), nrows.cuda(), traj)
    (cen * Gc.cuda()).sum().backward()
    for b in range(2):
        k = g[f"ids_{b}"].shape[0]
        torch.testing.assert_close(cen[b, :k].detach().cpu(), _t(g[f"centres_{b}"]), rtol=1e-4, atol=1e-5)
        dX = X.grad[b].cpu()
        scale = float(_t(g[f"dX_rownorm_{b}"]).max())
        big = float(_t(g[f"dX_ids_{b}"]).abs().max())
        torch.testing.assert_close(dX[ids[b, :k]], _t(g[f"dX_ids_{b}"]), rtol=0, atol=2e-3 * big)
        torch.testing.assert_close(dX[:64], _t(g[f"dX_head_{b}"]), rtol=0, atol=2e-3 * big)
        torch.testing.assert_close(dX.sum(0), _t(g[f"dX_colsum_{b}"]), rtol=0, atol=2e-3 * float(_t(g[f"dX_colsum_{b}"]).abs().max()))
        torch.testing.assert_close(dX.norm(dim=1), _t(g[f"dX_rownorm_{b}"]), rtol=0, atol=2e-3 * scale)
        assert abs(float(dX.norm()) - float(g[f"dX_norm_{b}"])) <= 1e-3 * float(g[f"dX_norm_{b}"])


@pytest.mark.parametrize("name", ["surface", "surface_cuboid", "volume", "volume_2", "volume_4"])
def test_unused_intersection_variants_by_reference_names(ref_names, golden, name):
    """compute_intersection_loss(_cuboid / _volume / _volume_2 / _volume_4) as upstream spells them: loss and the gradient
    with respect to every (r, V, c) against fit_intersections.npz (captured from the reference's autograd)."""
    CL = ref_names("convex_loss")
    g = golden("fit_intersections")
    P, surf, pts = intersection_case(g, "cuda")
    fn = {"surface": lambda: CL.compute_intersection_loss(P, surf),
          "surface_cuboid": lambda: CL.compute_intersection_loss_cuboid(P, surf),
          "volume": lambda: CL.compute_intersection_loss_volume(P, surf),
          "volume_2": lambda: CL.compute_intersection_loss_volume_2(P, pts),
          "volume_4": lambda: CL.compute_intersection_loss_volume_4(P, pts)}[name]
    loss = fn()
    assert loss.is_cuda
    torch.testing.assert_close(loss.detach().cpu().reshape(()), _t(g[f"{name}_loss"]), rtol=1e-4, atol=1e-8)
    loss.backward()
    check_intersection_grads(g, name, P, rtol=1e-3)


def test_sample_axis_and_empty_batches_by_reference_names(ref_names, golden):
    CL = ref_names("convex_loss")
    g = golden("fit_intersections")
    got = CL.sample_axis(_t(g["r_0"][1]).cuda(), _t(g["V_0"][1]).cuda(), _t(g["c_0"][1]).cuda())
    torch.testing.assert_close(got.cpu(), _t(g["axis_samples"]), rtol=1e-6, atol=1e-6)
    for fn in (CL.compute_intersection_loss, CL.compute_intersection_loss_volume):
        z = fn([], [])                                                 # convex_loss.py:158, :280: zeros(1) that needs grad
        assert z.shape == (1,) and z.item() == 0.0 and z.requires_grad and z.is_cuda
    labels = np.array([3, 0, 49, 7])
    hot = ref_names("src.ellipsoid_utils").to_one_hot(labels)          # src/ellipsoid_utils.py:146-154
    assert hot.shape == (4, 50) and hot.is_cuda and torch.equal(hot.argmax(1).cpu(), _t(labels)) and hot.sum().item() == 4
    one = [[(_t(g["r_2"][0]).cuda(), _t(g["V_2"][0]).cuda(), _t(g["c_2"][0]).cuda())]]
    pts = _t(g["pts"])[2:3].cuda()
    for fn in (CL.compute_intersection_loss_volume_2, CL.compute_intersection_loss_volume_4):
        z = fn(one, pts)                                               # a lone ellipsoid: every shape skipped (:356, :427)
        assert z.shape == (1,) and z.item() == 0.0 and z.requires_grad
