"""Generate tests/golden/*.npz from the upstream reference and validate the oracle against it.

TEST INFRASTRUCTURE -- runs only in the build container (needs /root/reference):

    python oracle/make_golden.py            # check oracle vs reference, then (re)write fixtures

A fixture is data only: seeds / small inputs and the reference's outputs.  Inputs that are large
are regenerated from a seed by `prifit_amd/synth.py` (the build's own generator) on both sides.
"""
import contextlib
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
import prifit_oracle as orc  # noqa: E402
from prifit_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@contextlib.contextmanager
def fixed_randint(values):
    """Make the reference's `torch.randint(0, N, (B,))` (pointnet_util.py:75) return chosen starts."""
    queue = [v.clone() for v in values]
    real = torch.randint

    def fake(*a, **k):
        return queue.pop(0)

    torch.randint = fake
    try:
        yield
    finally:
        torch.randint = real


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print("  wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def eq(a, b, what):
    assert torch.equal(a, b), "MISMATCH (exact) " + what
    print("  ok (bit-exact)  " + what)


def close(a, b, what, rtol=1e-5, atol=1e-6):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, "MISMATCH %s: max|diff|=%g (ref max %g)" % (what, err, ref)
    print("  ok (%.1e)      %s" % (err, what))


# ----------------------------------------------------------------------------------------------
def golden_index_ops():
    """FPS / ball query / 3-NN / square_distance on both synthetic distributions."""
    pu = refshim.ref("models.pointnet_util")
    print("[index ops]")
    for kind in ("cube", "surface"):
        for N, tag in ((2048, "n2048"), (1024, "n1024")):
            B = 4
            seed = {"cube": 1, "surface": 2}[kind] * 10 + (N // 1024)
            xyz = torch.from_numpy(synth.cloud(kind, B, N, seed))
            start1 = torch.from_numpy(synth.fps_start(B, N, seed))
            with fixed_randint([start1]):
                f1 = pu.farthest_point_sample(xyz, 512)
            eq(orc.farthest_point_sample(xyz, 512, start1), f1, f"fps torch {kind} {tag}")
            eq(orc.c_farthest_point_sample(xyz, 512, start1), f1, f"fps C     {kind} {tag}")
            c1 = pu.index_points(xyz, f1)
            start2 = torch.from_numpy(synth.fps_start(B, 512, seed + 100))
            with fixed_randint([start2]):
                f2 = pu.farthest_point_sample(c1, 128)
            eq(orc.c_farthest_point_sample(c1, 128, start2), f2, f"fps2 C    {kind} {tag}")
            c2 = pu.index_points(c1, f2)
            out = {"seed": seed, "start1": start1, "fps1": f1.to(torch.int16), "start2": start2,
                   "fps2": f2.to(torch.int16)}
            # MSG radii (sa1 on xyz/c1, sa2 on c1/c2) + SSG radii (cfg 1)
            cases = [("sa1", xyz, c1, [(0.1, 32), (0.2, 64), (0.4, 128), (0.2, 32)]),
                     ("sa2", c1, c2, [(0.4, 64), (0.8, 128), (0.4, 64)])]
            for lname, pts, ctr, rs in cases:
                for (r, k) in rs:
                    g = pu.query_ball_point(r, k, pts, ctr)
                    eq(orc.query_ball_point(r, k, pts, ctr), g, f"ball torch {kind} {tag} {lname} r={r}")
                    eq(orc.c_query_ball_point(r, k, pts, ctr), g, f"ball C     {kind} {tag} {lname} r={r}")
                    # store a checksum + a slice (full tensors are large): rows 0..63 of each shape
                    out[f"ball_{lname}_{r}_{k}_head"] = g[:, :64].to(torch.int16)
                    out[f"ball_{lname}_{r}_{k}_sum"] = g.sum(dim=(1, 2))
                    out[f"ball_{lname}_{r}_{k}_wsum"] = (g * (torch.arange(k) + 1)).sum(dim=(1, 2))
            # 3-NN as in fp1 (xyz <- c1) and fp2 (c1 <- c2)
            for lname, a, b in (("fp1", xyz, c1), ("fp2", c1, c2)):
                d = pu.square_distance(a, b)
                eq(orc.square_distance(a, b), d, f"sqdist torch {kind} {tag} {lname}")
                eq(orc.c_square_distance(a, b), d, f"sqdist C     {kind} {tag} {lname}")
                ds, ix = d.sort(dim=-1)
                ds, ix = ds[:, :, :3], ix[:, :, :3]
                d3, i3 = orc.c_three_nn(a, b)
                eq(i3, ix, f"3nn idx C {kind} {tag} {lname}")
                eq(d3, ds, f"3nn d   C {kind} {tag} {lname}")
                out[f"nn3_{lname}_idx"] = ix.to(torch.int16)
                out[f"nn3_{lname}_d"] = ds
            out["sqdist_fp2_head"] = pu.square_distance(c1, c2)[:2, :128]
            save(f"index_{kind}_{tag}", **out)


def copy_state(dst, src):
    missing = dst.load_state_dict(src.state_dict(), strict=True)
    return missing


def golden_modules():
    """SA-MSG / SA(group_all) / FP modules: fwd outputs + grads with shared seeded parameters."""
    pu = refshim.ref("models.pointnet_util")
    print("[modules]")
    B, N = 2, 512
    seed = 5
    xyz = torch.from_numpy(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    feat = torch.from_numpy(synth.features(B, N, 16, seed)).transpose(1, 2).contiguous()
    start = torch.from_numpy(synth.fps_start(B, N, seed))

    torch.manual_seed(11)
    ref_sa = pu.PointNetSetAbstractionMsg(64, [0.2, 0.4], [8, 16], 16, [[16, 32], [16, 24, 32]])
    synth.perturb_bn(ref_sa, 3)
    my_sa = orc.OracleSetAbstractionMsg(64, [0.2, 0.4], [8, 16], 16, [[16, 32], [16, 24, 32]])
    copy_state(my_sa, ref_sa)
    f_ref = feat.clone().requires_grad_(True)
    with fixed_randint([start]):
        rx, rp = ref_sa(xyz, f_ref)
    gout = torch.from_numpy(synth.features(B, 64, rp.shape[1], seed + 1)).transpose(1, 2)
    (rp * gout).sum().backward()
    f_my = feat.clone().requires_grad_(True)
    mx, mp = my_sa(xyz, f_my, start)
    (mp * gout).sum().backward()
    eq(mx, rx, "sa_msg new_xyz")
    close(mp, rp, "sa_msg out")
    close(f_my.grad, f_ref.grad, "sa_msg dfeat", rtol=1e-4)
    gw = {k: p.grad for k, p in ref_sa.named_parameters()}
    # conv biases feed a train-mode BN, so their true gradient is 0 and what is left is rounding
    # noise: use an absolute tolerance scaled by the largest gradient of the module.
    gmax = max(v.abs().max().item() for v in gw.values())
    for k, p in my_sa.named_parameters():
        close(p.grad, gw[k], "sa_msg d" + k, rtol=2e-4, atol=2e-5 * gmax)
    save("module_sa_msg", seed=seed, start=start, new_xyz=rx, out=rp, dfeat=f_ref.grad,
         running_mean_00=ref_sa.bn_blocks[0][0].running_mean, running_var_00=ref_sa.bn_blocks[0][0].running_var,
         **{"g_" + k: v for k, v in gw.items()})

    # group_all SA
    torch.manual_seed(12)
    ref_ga = pu.PointNetSetAbstraction(None, None, None, 16 + 3, [32, 64], True)
    synth.perturb_bn(ref_ga, 4)
    my_ga = orc.OracleSetAbstraction(None, None, None, 16 + 3, [32, 64], True)
    copy_state(my_ga, ref_ga)
    f_ref = feat.clone().requires_grad_(True)
    _, rp = ref_ga(xyz, f_ref)
    g2 = torch.from_numpy(synth.features(B, 1, 64, seed + 2)).transpose(1, 2)
    (rp * g2).sum().backward()
    f_my = feat.clone().requires_grad_(True)
    _, mp = my_ga(xyz, f_my)
    (mp * g2).sum().backward()
    close(mp, rp, "sa_all out")
    close(f_my.grad, f_ref.grad, "sa_all dfeat", rtol=1e-4)
    save("module_sa_all", seed=seed, out=rp, dfeat=f_ref.grad,
         **{"g_" + k: p.grad for k, p in ref_ga.named_parameters()})

    # SSG SA (cfg 1 concat order)
    torch.manual_seed(13)
    ref_ss = pu.PointNetSetAbstraction(64, 0.3, 16, 16 + 3, [32, 64], False)
    synth.perturb_bn(ref_ss, 5)
    my_ss = orc.OracleSetAbstraction(64, 0.3, 16, 16 + 3, [32, 64], False)
    copy_state(my_ss, ref_ss)
    f_ref = feat.clone().requires_grad_(True)
    with fixed_randint([start]):
        rx, rp = ref_ss(xyz, f_ref)
    g3 = torch.from_numpy(synth.features(B, 64, 64, seed + 3)).transpose(1, 2)
    (rp * g3).sum().backward()
    f_my = feat.clone().requires_grad_(True)
    mx, mp = my_ss(xyz, f_my, start)
    (mp * g3).sum().backward()
    eq(mx, rx, "sa_ssg new_xyz")
    close(mp, rp, "sa_ssg out")
    close(f_my.grad, f_ref.grad, "sa_ssg dfeat", rtol=1e-4)
    save("module_sa_ssg", seed=seed, start=start, out=rp, dfeat=f_ref.grad,
         **{"g_" + k: p.grad for k, p in ref_ss.named_parameters()})

    # FP (3-NN branch and S==1 branch)
    torch.manual_seed(14)
    S = 64
    xyz2 = xyz[:, :, :S].contiguous()
    p1 = torch.from_numpy(synth.features(B, N, 8, seed + 4)).transpose(1, 2).contiguous()
    p2 = torch.from_numpy(synth.features(B, S, 24, seed + 5)).transpose(1, 2).contiguous()
    ref_fp = pu.PointNetFeaturePropagation(32, [32, 16])
    synth.perturb_bn(ref_fp, 6)
    my_fp = orc.OracleFeaturePropagation(32, [32, 16])
    copy_state(my_fp, ref_fp)
    a1, a2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    ro = ref_fp(xyz, xyz2, a1, a2)
    g4 = torch.from_numpy(synth.features(B, N, 16, seed + 6)).transpose(1, 2)
    (ro * g4).sum().backward()
    b1, b2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    mo = my_fp(xyz, xyz2, b1, b2)
    (mo * g4).sum().backward()
    close(mo, ro, "fp out")
    close(b1.grad, a1.grad, "fp dpoints1", rtol=1e-4)
    close(b2.grad, a2.grad, "fp dpoints2", rtol=1e-4)
    save("module_fp", seed=seed, out=ro, dpoints1=a1.grad, dpoints2=a2.grad,
         **{"g_" + k: p.grad for k, p in ref_fp.named_parameters()})

    p2g = p2[:, :, :1].contiguous()
    a1, a2 = p1.clone().requires_grad_(True), p2g.clone().requires_grad_(True)
    ro = ref_fp(xyz, xyz2[:, :, :1], a1, a2)
    (ro * g4).sum().backward()
    b1, b2 = p1.clone().requires_grad_(True), p2g.clone().requires_grad_(True)
    mo = my_fp(xyz, xyz2[:, :, :1], b1, b2)
    (mo * g4).sum().backward()
    close(mo, ro, "fp(S=1) out")
    close(b2.grad, a2.grad, "fp(S=1) dpoints2", rtol=1e-4)
    save("module_fp_s1", seed=seed, out=ro, dpoints2=a2.grad)


def _module_summary(prefix, out, dfeat, grads):
    """Small, tight summary of a big module run: heads, sums along both axes, one seeded random projection."""
    d = {prefix + "out_head": out[:, :, :8].detach(), prefix + "out_sum_s": out.detach().sum(2),
         prefix + "out_sum_c": out.detach().sum(1)}
    if dfeat is not None:
        d[prefix + "dfeat_head"] = dfeat[:, :, :16].detach()
        d[prefix + "dfeat_sum_n"] = dfeat.detach().sum(2)
        d[prefix + "dfeat_sum_c"] = dfeat.detach().sum(1)
    for k, v in grads.items():
        d[prefix + "g_" + k] = v
    return d


def _fp64_noise(my_mod, xyz, feat, start, gout, ref_out, ref_grads, ref_dfeat, what):
    """fp32 irreproducibility of the module's gradients: the oracle module in float64 (same parameters, same inputs,
    same indices -- checked through new_xyz and the output) against the REFERENCE's fp32 run.  A 1e-6 forward
    difference flips a max-pool winner / ReLU mask now and then, so fp32 gradients of these sums over ~10^6 grouped
    rows are only defined to ~1e-3; the per-parameter relative L2 deviation measured here is stored in the fixture and
    is what bounds the tolerance of the GPU test (max(2e-4, 2 x noise))."""
    import copy
    m64 = copy.deepcopy(my_mod).double()
    for p in m64.parameters():
        p.grad = None
    f64 = feat.double().clone().requires_grad_(True)
    nx, out = m64(xyz.double(), f64, start)
    (out * gout.double()).sum().backward()
    err = (out.float() - ref_out).abs().max().item()
    assert err < 2e-5 * max(1.0, ref_out.abs().max().item()), "fp64 run took different indices? " + what
    noise = {k: ((p.grad.float() - ref_grads[k]).norm() / ref_grads[k].norm().clamp(min=1e-30)).item()
             for k, p in m64.named_parameters()}
    dnoise = ((f64.grad.float() - ref_dfeat).norm() / ref_dfeat.norm()).item()
    print("  fp32-vs-fp64 noise %s: params max %.1e (%s), dfeat %.1e" %
          (what, max(noise.values()), max(noise, key=noise.get), dnoise))
    return noise, dnoise, {k: p.grad.float() for k, p in m64.named_parameters()}


def _module_tolerance(mine, ref, noise, my_dfeat, ref_dfeat, dnoise, what):
    """Relative-L2 bound for the module's gradient tensors = 2 x the largest deviation measured between three
    evaluations of the SAME arithmetic (reference fp32, oracle fp32, oracle fp64), floor 2e-4.  The deviations are
    sparse random events (a ReLU mask / max-pool winner flipping on a 1e-6 forward difference): which tensor a given
    run's flips land on differs from run to run, so the bound is per module, not per tensor."""
    gmax = max(v.abs().max().item() for v in ref.values())
    ddev = max(dnoise, ((my_dfeat - ref_dfeat).norm() / ref_dfeat.norm()).item())
    devs = []
    for k, g in mine.items():
        if k.endswith(".bias") and "conv" in k:
            # bias in front of a train-mode BatchNorm: true gradient 0, the value is rounding noise
            assert g.abs().max().item() <= 1e-3 * gmax, (what, k)
            continue
        devs += [noise[k], ((g - ref[k]).norm() / ref[k].norm()).item()]
        if os.environ.get("GOLDEN_VERBOSE"):
            print("      %-28s |g| %.3e  fp64-vs-ref %.1e  oracle-vs-ref %.1e" % (k, ref[k].norm().item(), devs[-2], devs[-1]))
    worst = max(devs)
    assert max(worst, ddev) < 1e-2, "MISMATCH %s: gradients deviate by %.2e / %.2e (more than fp32 noise can explain)" % (what, worst, ddev)
    tol, dtol = max(2e-4, 2.0 * worst), max(2e-4, 2.0 * ddev)
    print("  ok  %s gradients: largest deviation across reference-fp32 / oracle-fp32 / oracle-fp64: parameters %.1e -> tol "
          "%.1e, input features %.1e -> tol %.1e" % (what, worst, tol, ddev, dtol))
    return tol, dtol


def golden_modules_real():
    """The ACTUAL set-abstraction levels of the MSG part-seg network (models/pointnet2_part_seg_msg.py:27-28,
    models/pointnet_util.py:223-261) at their real shapes: SA1 (512, [.1,.2,.4], [32,64,128], 3, ...) on B=4 x 2048
    surface points (l0_points = xyz), SA2 (128, [.4,.8], [64,128], 320, ...) on B=4 x 512 points with 320 features.
    Outputs, input-feature gradients and EVERY parameter gradient; these shapes reach the streaming GEMMs, the
    SA1 direct mode and the fused BatchNorm / max-pool epilogues of the HIP backend."""
    pu = refshim.ref("models.pointnet_util")
    print("[modules, real shapes]")
    B, seed = 4, 61
    out = {"seed": seed}
    cases = (("sa1", 2048, None, (512, [0.1, 0.2, 0.4], [32, 64, 128], 3, [[32, 32, 64], [64, 64, 128], [64, 96, 128]]), 61, 13),
             ("sa2", 512, 320, (128, [0.4, 0.8], [64, 128], 128 + 128 + 64, [[128, 128, 256], [128, 196, 256]]), 62, 14))
    for i, (name, N, C, cfg, tseed, bnseed) in enumerate(cases):
        xyz = torch.from_numpy(synth.cloud("surface", B, N, seed + 2 * i)).transpose(1, 2).contiguous()
        feat = xyz if C is None else torch.from_numpy(synth.features(B, N, C, seed + 2 * i + 1)).transpose(1, 2).contiguous()
        start = torch.from_numpy(synth.fps_start(B, N, seed + 2 * i))
        torch.manual_seed(tseed)
        ref_sa = pu.PointNetSetAbstractionMsg(*cfg)
        synth.xavier_like_trainer(ref_sa)
        synth.perturb_bn(ref_sa, bnseed)
        my_sa = orc.OracleSetAbstractionMsg(*cfg)
        copy_state(my_sa, ref_sa)
        f_ref = feat.clone().requires_grad_(True)
        with fixed_randint([start]):
            rx, rp = ref_sa(xyz, f_ref)
        gout = torch.from_numpy(synth.features(B, cfg[0], rp.shape[1], seed + 10 + i)).transpose(1, 2)
        (rp * gout).sum().backward()
        gw = {k: p.grad for k, p in ref_sa.named_parameters()}
        f_my = feat.clone().requires_grad_(True)
        mx, mp = my_sa(xyz, f_my, start)
        (mp * gout).sum().backward()
        eq(mx, rx, name + " new_xyz")
        close(mp, rp, name + " out", rtol=1e-5, atol=1e-5)
        noise, dnoise, g64 = _fp64_noise(my_sa, xyz, feat, start, gout, rp.detach(), gw, f_ref.grad, name)
        tol, dtol = _module_tolerance({k: p.grad for k, p in my_sa.named_parameters()}, gw, noise, f_my.grad, f_ref.grad,
                                dnoise, name)
        out[name + "_grad_tol"] = np.array(tol)
        out[name + "_dfeat_tol"] = np.array(dtol)
        out[name + "_start"] = start
        out[name + "_new_xyz"] = rx
        out.update(_module_summary(name + "_", rp, f_ref.grad, gw))
        last = ref_sa.bn_blocks[-1][-1]
        out[name + "_running_mean_last"] = last.running_mean
        out[name + "_running_var_last"] = last.running_var
    save("module_sa_real", **out)


def golden_selfsup_step():
    """the self-supervised training iteration of train_partseg_shapenet.py:436-451 through
    the MAIN network file (models/pointnet2_part_seg_msg.py:64-134, include_convex_loss=True): zero_grad, train(),
    forward on a 2048-subset of the 5000 chamfer points, mean(loss) * lambda, backward, Adam step (train:252-259).
    Harness conventions as in make_golden_fit.py (shared covariance noise, pinned SVD signs, Fibonacci sampler), plus
    one this step needs: the REPRESENTATIVE ids of the modes.  `center = new_X[indices]` (src/mean_shift.py:46) is a
    differentiable gather, so the loss gradient enters the mean-shift trajectory of exactly the point nms picked to
    represent a collapsed mode -- and that pick is last-bit noise.  Measured here, everything else equal:
    oracle with its own picks vs the reference: d loss / d embedding differs by 12 % (relative L2), parameter-gradient
    norms by -5 ... +14 %; with the reference's picks passed in (`center_ids`): 9e-5 and < 1e-3.  The fixture stores the
    picks; forward quantities (loss, K, partition) do not depend on them."""
    import make_golden_fit as F_
    print("[self-supervised step]")
    M = refshim.ref("models.pointnet2_part_seg_msg")
    EF = refshim.ref("src.ellipsoid_fitting")
    SE = refshim.ref("src.sample_ellipsoid")
    B, N, seed = 2, 2048, 71
    cham_np, lab_np = synth.blobs_with_labels(B, 5000, seed)
    cham = torch.from_numpy(cham_np)
    sel = torch.from_numpy(np.random.default_rng(seed + 1).choice(5000, N, replace=False))
    pts = cham[:, sel]
    xyz = pts.transpose(1, 2).contiguous()
    cham_t = cham.transpose(1, 2).contiguous()
    cls = torch.zeros(B, 1, 16)
    s1 = torch.from_numpy(synth.fps_start(B, N, seed))
    s2 = torch.from_numpy(synth.fps_start(B, 512, seed + 100))
    R = torch.from_numpy(synth.uniform01((3, 3), seed))
    q, iters = 0.05, 10
    torch.manual_seed(24)
    ref_net = M.get_model(50)
    synth.xavier_like_trainer(ref_net)
    synth.perturb_bn(ref_net, 9)
    # A seeded, UNTRAINED network maps every point of a shape to nearly the same embedding direction: mean-shift then
    # finds one cluster and every gradient of the step is rounding noise (measured: K = [1, 1], |dW| 3e-8).  The
    # embedding head `extra_conv_emb` (a 128 x 128 linear map + bias) is therefore PRE-CONDITIONED once, in closed form:
    # ridge regression (lambda = 1, float64) from the reference's own `feat` on this batch to a random unit prototype per
    # generating blob.  The embedding then separates the 8 blobs softly (mean cosine to the own prototype 0.94): K = 8
    # with the same partition in fp32 and fp64.  The fitted head is part of the fixture.
    import copy
    pre = copy.deepcopy(ref_net).train()
    pre.drop1.eval()
    cap = {}
    pre.bn1.register_forward_hook(lambda m, i, o: cap.__setitem__("y", o))
    with torch.no_grad(), fixed_randint([s1, s2]):
        try:
            pre(xyz, cls)
        except UnboundLocalError:          # upstream's return statement needs the convex loss: feat is captured
            pass
    feat0 = torch.relu(cap["y"]).double()                                  # feat [B,128,N] (msg:88)
    A1 = torch.cat([feat0.permute(0, 2, 1).reshape(-1, 128), torch.ones(B * N, 1, dtype=torch.float64)], 1)
    proto = np.random.default_rng(seed + 5).normal(size=(8, 128))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    P = torch.from_numpy(proto)[torch.from_numpy(lab_np)[:, sel].reshape(-1)]
    sol = torch.linalg.solve(A1.T @ A1 + 1.0 * torch.eye(129, dtype=torch.float64), A1.T @ P)
    emb_W, emb_b = sol[:128].T.float().contiguous(), sol[128].float().contiguous()
    with torch.no_grad():
        ref_net.extra_conv_emb.weight.copy_(emb_W.unsqueeze(-1))
        ref_net.extra_conv_emb.bias.copy_(emb_b)
    my_net = orc.OracleMSGPartSeg(50)
    copy_state(my_net, ref_net)
    for net in (ref_net, my_net):
        net.train()
        net.drop1.eval()

    def ref_customsvd_canonical(Mx):
        U, S, V = refshim.ref("src.fitting_utils").customsvd(Mx)
        sg = orc.canonical_signs(V).view(1, 3)
        return U * sg, S, V * sg

    def ref_sample(self, a, b_, c, center, transformation, n=500):
        U, V = orc.fibonacci_uv(int(n))
        p = self.uniform_sample_points_on_ellipsoid(U, V, a, b_, c)
        return p @ transformation.T + center, None

    # record which point the reference's nms picks to represent each mode (rounding noise, yet the gradient
    # enters through exactly that point's mean-shift trajectory)
    MS = refshim.ref("src.mean_shift")
    real_nms = MS.MeanShift.nms
    ref_ids = []

    def recording_nms(self, centers, X, b):
        out3 = real_nms(self, centers, X, b)
        ref_ids.append(out3[1].clone())
        return out3

    opt_r = torch.optim.Adam(ref_net.parameters(), lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    opt_r.zero_grad()
    with fixed_randint([s1, s2]), F_.patched(torch, "rand", lambda *a, **k: R.clone()), \
            F_.patched(MS.MeanShift, "nms", recording_nms), \
            F_.patched(EF, "customsvd", ref_customsvd_canonical), F_.patched(SE.SampleEllipsoid, "sample", ref_sample):
        r_out = ref_net(xyz, cls, chamfer_points=cham_t, include_convex_loss=True, quantile=q, msc_iterations=iters,
                        max_num_clusters=25)
    _, _, rfeat, rtot, rcham, rlabels, rparams, remb = r_out
    remb.retain_grad()
    rfeat.retain_grad()
    if os.environ.get("GOLDEN_DEBUG"):
        for prm in rparams:
            for t3 in prm:
                for t in t3:
                    t.retain_grad()
    (torch.mean(rtot) * 1.0).backward()
    rg = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in ref_net.named_parameters()}
    before = {k: p.detach().clone() for k, p in ref_net.named_parameters()}
    opt_r.step()

    opt_o = torch.optim.Adam(my_net.parameters(), lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    opt_o.zero_grad()
    assert len(ref_ids) == B, "a quantile-doubling retry happened: pick other inputs"
    o_out = my_net(xyz, cls, chamfer_points=cham_t, include_convex_loss=True, quantile=q, msc_iterations=iters,
                   max_num_clusters=25, fps_start=(s1, s2),
                   fit_inputs=dict(rand_table=[[R] * 64] * B, canonical=True, center_ids=ref_ids))
    _, _, ofeat, otot, ocham, olabels, oparams, oemb = o_out
    oemb.retain_grad()
    ofeat.retain_grad()
    if os.environ.get("GOLDEN_DEBUG"):
        for prm in oparams:
            for t3 in prm:
                for t in t3:
                    t.retain_grad()
    torch.mean(otot).backward()
    if os.environ.get("GOLDEN_DEBUG"):
        for b in range(B):
            order_r = F_.canonical_order(rlabels[b], len(rparams[b]))
            order_o = F_.canonical_order(olabels[b], len(oparams[b]))
            for kr, ko in zip(order_r.tolist(), order_o.tolist()):
                pr, po = rparams[b][kr], oparams[b][ko]
                print("   b%d cluster %d/%d: r %s | %s  dr %s | %s  dc %s | %s" % (b, kr, ko, pr[0].detach().numpy().round(4), po[0].detach().numpy().round(4),
                      pr[0].grad.numpy().round(5), po[0].grad.numpy().round(5), pr[2].grad.numpy().round(5), po[2].grad.numpy().round(5)))
    print("  dX (embedding gradient) oracle-vs-reference rel L2 %.2e; dfeat rel L2 %.2e" %
          (((oemb.grad - remb.grad).norm() / remb.grad.norm()).item(), ((ofeat.grad - rfeat.grad).norm() / rfeat.grad.norm()).item()))
    og = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in my_net.named_parameters()}
    opt_o.step()

    assert abs(ref_net.beta - 0.99) < 1e-12 and abs(my_net.beta - 0.99) < 1e-12
    close(ofeat, rfeat, "selfsup feat", rtol=1e-4, atol=1e-5)
    close(oemb, remb, "selfsup embedding", rtol=1e-4, atol=1e-5)
    Ks = [len(p) for p in rparams]
    assert Ks == [len(p) for p in oparams], (Ks, [len(p) for p in oparams])
    for b in range(B):
        assert F_.same_partition(rlabels[b], olabels[b]), "label partition differs b=%d" % b
    print("  ok partition      K per shape", Ks)
    close(otot, rtot, "selfsup total_loss", rtol=1e-4)
    close(ocham, rcham, "selfsup chamfer_loss", rtol=1e-4)
    names = [k for k in rg if rg[k] is not None]
    assert "conv2.weight" not in names and "extra_conv_emb.weight" in names      # seg head is off this path
    gn_r = {k: rg[k].norm().item() for k in names}
    rels = {}
    for k in names:
        if k.endswith(".bias") and "conv" in k and k != "extra_conv_emb.bias":
            continue      # bias in front of a train-mode BatchNorm: true gradient 0
        rel = ((og[k] - rg[k]).norm() / max(rg[k].norm().item(), 1e-30)).item()
        print("    grad %-32s |g| %.3e  oracle-vs-reference rel %.1e  norm ratio %.4f" % (k, gn_r[k], rel, og[k].norm().item() / gn_r[k]))
        rels[k] = rel
        assert rel < 2e-2, (k, rel)     # with the representative ids pinned; unpinned: 0.06 - 0.33 (see the docstring)
    # parameter checksum after the Adam step: the first Adam step moves every entry by lr*g/(|g|+eps) ~ lr*sign(g), so
    # entries whose gradient is rounding noise move by +-lr at random; the checksum is the update's L2 norm per
    # parameter (insensitive to those signs) plus the full post-step tensors of two key parameters
    upd_norm = {k: (p.detach() - before[k]).norm().item() for k, p in ref_net.named_parameters()}
    save("step_selfsup", seed=seed, R=R, s1=s1, s2=s2, total_loss=rtot.detach(), chamfer_loss=rcham.detach(),
         K=np.array(Ks), labels=torch.stack(rlabels).to(torch.int16), beta=np.array(ref_net.beta),
         feat_head=rfeat[:, :, :64].detach(), emb_head=remb[:, :, :64].detach(),
         grad_names=np.array(names), grad_norms=np.array([gn_r[k] for k in names]),
         g_extra_conv_emb_weight=rg["extra_conv_emb.weight"], g_conv1_weight=rg["conv1.weight"],
         g_sa1_first=rg["sa1.conv_blocks.0.0.weight"],
         upd_names=np.array(sorted(upd_norm)), upd_norms=np.array([upd_norm[k] for k in sorted(upd_norm)]),
         p_extra_conv_emb_weight=dict(ref_net.named_parameters())["extra_conv_emb.weight"].detach(),
         emb_W=emb_W, emb_b=emb_b,
         center_ids=torch.stack([torch.cat([i, torch.full((32 - i.shape[0],), -1, dtype=torch.long)]) for i in ref_ids]).to(torch.int16),
         g_emb_head=remb.grad[:, :, :32].contiguous(), g_emb_norm=remb.grad.norm())


def golden_model():
    """Full MSG part-seg network, B=2 x 2048, supervised step (train_partseg_shapenet.py:382-399)."""
    print("[model]")
    M = refshim.ref("models.pretrain_pointnet2_part_seg_msg")
    B, N = 2, 2048
    seed = 7
    torch.manual_seed(21)
    ref_net = M.get_model(50)
    synth.xavier_like_trainer(ref_net)
    synth.perturb_bn(ref_net, 8)
    my_net = orc.OracleMSGPartSeg(50)
    copy_state(my_net, ref_net)
    for net in (ref_net, my_net):
        net.train()
        net.drop1.eval()  # compare with dropout off, BN in train mode
    xyz = torch.from_numpy(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    cls = torch.zeros(B, 1, 16)
    cls[:, 0, 3] = 1.0
    target = torch.from_numpy(synth.labels(B, N, 50, seed))
    s1 = torch.from_numpy(synth.fps_start(B, N, seed))
    s2 = torch.from_numpy(synth.fps_start(B, 512, seed + 100))
    with fixed_randint([s1, s2]):
        rseg, (rl1, rl2, rl3), rfeat, _, _ = ref_net(xyz, cls)
    rloss = M.get_loss()(rseg.contiguous().view(-1, 50), target.view(-1), None)
    rloss.backward()
    mseg, (ml1, ml2, ml3), mfeat, _, _ = my_net(xyz, cls, fps_start=(s1, s2))
    mloss = orc.seg_loss(mseg.contiguous().view(-1, 50), target.view(-1))
    mloss.backward()
    close(mseg, rseg, "model seg log-probs", rtol=1e-4, atol=1e-5)
    close(mfeat, rfeat, "model feat", rtol=1e-4, atol=1e-5)
    close(ml3, rl3, "model l3", rtol=1e-4, atol=1e-5)
    close(mloss, rloss, "model CE loss", rtol=1e-5)
    rg = {k: p.grad for k, p in ref_net.named_parameters()}
    gmax = max(v.abs().max().item() for v in rg.values() if v is not None)
    gn = {}
    for k, p in my_net.named_parameters():
        if rg[k] is None:
            assert p.grad is None or p.grad.abs().max() == 0, k
            continue
        gn[k] = rg[k].norm().item()
        # Whole-network gradients are only reproducible to ~1e-2 in fp32: 1e-5 forward differences
        # (summation order) flip a few ReLU masks / max-pool winners, and the CE gradient sums cancel
        # heavily.  Measured here: reference-fp32 vs an fp64 run of this oracle differs by 2.5e-3 on
        # sa1.conv_blocks.0.0.weight.  Per-module fixtures (module_*.npz) carry the tight tolerances.
        if k.endswith(".bias") and ("conv" in k) and k != "conv2.bias":
            continue  # bias in front of a train-mode BN: true gradient is 0, value is noise
        rel = ((p.grad - rg[k]).norm() / rg[k].norm()).item()
        assert rel < 3e-2, (k, rel)
    # fp32 irreproducibility of the whole-network gradients, MEASURED: the oracle in float64 (same parameters, inputs,
    # indices) against the reference's fp32 run, per parameter -- relative deviation of the norm and relative L2 of the
    # vector.  The GPU test's bars are 2 x the largest of {this, oracle-fp32 vs reference} (model_msg_sup_noise.npz).
    import copy
    net64 = copy.deepcopy(my_net).double()
    for p in net64.parameters():
        p.grad = None
    seg64 = net64(xyz.double(), cls.double(), fps_start=(s1, s2))[0]
    assert (seg64.float() - rseg).abs().max().item() < 1e-3, "fp64 run took other indices"
    orc.seg_loss(seg64.contiguous().view(-1, 50), target.view(-1)).backward()
    dn, dv = [], []
    for (k, p), (_, p64) in zip(my_net.named_parameters(), net64.named_parameters()):
        if rg[k] is None or (k.endswith(".bias") and ("conv" in k) and k != "conv2.bias"):
            continue
        for other in (p.grad, p64.grad.float()):
            dn.append(abs(other.norm().item() - rg[k].norm().item()) / rg[k].norm().item())
            dv.append(((other - rg[k]).norm() / rg[k].norm()).item())
    print("  whole-network gradients, largest deviation across reference-fp32 / oracle-fp32 / oracle-fp64: norm %.1e, "
          "vector %.1e" % (max(dn), max(dv)))
    save("model_msg_sup_noise", norm_dev=np.array(max(dn)), vec_dev=np.array(max(dv)),
         norm_tol=np.array(max(2e-3, 2 * max(dn))), vec_tol=np.array(max(2e-3, 2 * max(dv))))
    names = sorted(gn)
    save("model_msg_sup", seed=seed, s1=s1, s2=s2, loss=rloss.detach(), seg_head=rseg[:, :64].detach(),
         seg_sum=rseg.detach().sum(dim=1), feat_head=rfeat[:, :, :64].detach(), l3=rl3.detach(),
         l2_head=rl2[:, :, :16].detach(), grad_names=np.array(names), grad_norms=np.array([gn[k] for k in names]),
         g_conv2_weight=rg["conv2.weight"], g_sa1_first=rg["sa1.conv_blocks.0.0.weight"],
         g_fp1_last=rg["fp1.mlp_convs.1.weight"])

    # cfg 1: SSG plumbing case, 4 x 1024 (models/pointnet2_part_seg_ssg.py)
    S = refshim.ref("models.pointnet2_part_seg_ssg")
    torch.manual_seed(22)
    ssg = S.get_model(50)
    ssg.train()
    ssg.drop1.eval()
    xyz4 = torch.from_numpy(synth.cloud("cube", 4, 1024, 9)).transpose(1, 2).contiguous()
    t1 = torch.from_numpy(synth.fps_start(4, 1024, 9))
    t2 = torch.from_numpy(synth.fps_start(4, 512, 109))
    with fixed_randint([t1, t2]):
        sseg, sl3 = ssg(xyz4, torch.zeros(4, 1, 16))
    torch.manual_seed(22)  # same construction order => same seeded init (checked here)
    my_ssg = orc.OracleSSGPartSeg(50)
    for (ka, va), (kb, vb) in zip(ssg.named_parameters(), my_ssg.named_parameters()):
        assert ka == kb and torch.equal(va, vb), ka
    my_ssg.train()
    my_ssg.drop1.eval()
    oseg, ol3 = my_ssg(xyz4, torch.zeros(4, 1, 16), fps_start=(t1, t2))
    close(oseg, sseg, "ssg seg log-probs", rtol=1e-4, atol=1e-5)
    close(ol3, sl3, "ssg l3", rtol=1e-4, atol=1e-5)
    save("model_ssg", seed=9, s1=t1, s2=t2, seg_sum=sseg.detach().sum(dim=1), l3=sl3.detach())


def golden_variants():
    """cls MSG / cls SSG / sem-seg variants (models/pointnet2_cls_msg.py, pointnet2_cls_ssg.py, pointnet2_sem_seg.py)
    on 2 x 1024 points: oracle vs reference forward with identical seeded init and FPS starts."""
    print("[variants]")
    sys.path.insert(0, os.path.join(refshim.REF_ROOT, "models"))   # the cls files import pointnet_util bare
    xyz = torch.from_numpy(synth.cloud("surface", 2, 1024, 12)).transpose(1, 2).contiguous()
    starts = [torch.from_numpy(synth.fps_start(2, n, 200 + i)) for i, n in enumerate((1024, 512, 256, 64))]
    out = {"seed": 12}
    for name, modname, kw, okw in (("cls_msg", "pointnet2_cls_msg", dict(normal_channel=False), dict(msg=True)),
                                   ("cls_ssg", "pointnet2_cls_ssg", dict(normal_channel=False), dict(msg=False))):
        R = importlib.import_module(modname)
        torch.manual_seed(5)
        ref = R.get_model(40, **kw)
        ref.train(); ref.drop1.eval(); ref.drop2.eval()
        with fixed_randint(starts[:2]):
            lp, l3 = ref(xyz)
        torch.manual_seed(5)
        mine = orc.OracleCls(40, normal_channel=False, **okw)
        for (ka, va), (kb, vb) in zip(ref.named_parameters(), mine.named_parameters()):
            assert ka == kb and torch.equal(va, vb), ka
        mine.train(); mine.drop1.eval(); mine.drop2.eval()
        olp, ol3 = mine(xyz, fps_start=(starts[0], starts[1]))
        close(olp, lp, name + " log-probs", rtol=1e-4, atol=1e-5)
        close(ol3, l3, name + " l3", rtol=1e-4, atol=1e-5)
        out[name + "_logp"] = lp.detach()
        out[name + "_l3_sum"] = l3.detach().sum(dim=1)
    SS = refshim.ref("models.pointnet2_sem_seg")
    feats = torch.from_numpy(synth.features(2, 1024, 3, 13)).transpose(1, 2)   # with_rgb: input is xyz + rgb
    x9 = torch.cat([xyz, feats], 1).contiguous()
    torch.manual_seed(6)
    ref = SS.get_model(13, with_rgb=True)
    ref.train(); ref.drop1.eval()
    with fixed_randint(starts):
        lp, l4 = ref(x9)
    torch.manual_seed(6)
    mine = orc.OracleSemSeg(13, with_rgb=True)
    for (ka, va), (kb, vb) in zip(ref.named_parameters(), mine.named_parameters()):
        assert ka == kb and torch.equal(va, vb), ka
    mine.train(); mine.drop1.eval()
    olp, ol4 = mine(x9, fps_start=tuple(starts))
    close(olp, lp, "sem_seg log-probs", rtol=1e-4, atol=1e-5)
    close(ol4, l4, "sem_seg l4", rtol=1e-4, atol=1e-5)
    out["sem_logp_sum"] = lp.detach().sum(dim=1)
    out["sem_l4"] = l4.detach()
    for i, t in enumerate(starts):
        out["s%d" % i] = t
    save("model_variants", **out)


class _TorchProxy:  # src/dgcnn.py:83,122 hard-code torch.device('cuda')
    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def device(*a, **k):
        return torch.device("cpu")


def _dgcnn_pair(num_channels, k, seed_w=31):
    """(reference DGCNGn, oracle OracleDGCNGn) with the same seeded parameters; GroupNorm affine terms off their defaults."""
    D = refshim.ref("src.dgcnn")
    D.torch = _TorchProxy()
    torch.manual_seed(seed_w)
    ref = D.DGCNGn(emb_size=128, num_channels=num_channels, nn_nb=k)
    for m in ref.modules():
        if isinstance(m, torch.nn.GroupNorm):
            with torch.no_grad():
                g = torch.Generator().manual_seed(m.num_channels)
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
    my = orc.OracleDGCNGn(128, num_channels, k)
    assert [kk for kk, _ in my.state_dict().items()] == [kk for kk, _ in ref.state_dict().items()]
    my.load_state_dict(ref.state_dict())
    return D, ref, my


def dgcnn_normals_input(B, N, seed):
    """[B,6,N]: surface cloud + unit pseudo-normals (the build's generator on both sides)."""
    pts = torch.from_numpy(synth.cloud("surface", B, N, seed))
    nrm = torch.from_numpy(synth.features(B, N, 3, seed + 7))
    nrm = nrm / nrm.norm(dim=2, keepdim=True)
    return torch.cat([pts, nrm], dim=2).transpose(1, 2).contiguous()


def _golden_dgcnn_case(name, B, N, k, seed, num_channels=3):
    D, ref, my = _dgcnn_pair(num_channels, k)
    if num_channels == 6:
        pts = dgcnn_normals_input(B, N, seed)
        idx_r = D.knn_points_normals(pts, k, k)
        eq(orc.knn_points_normals(pts, k, k), idx_r, "dgcnn knn_points_normals")
    else:
        pts = torch.from_numpy(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
        idx_r = D.knn(pts, k, k)
        eq(orc.knn(pts, k, k), idx_r, "dgcnn knn (xyz)")
    ge = torch.from_numpy(synth.features(B, N, 128, seed + 1))
    gs = torch.from_numpy(synth.features(B, N, 3, seed + 2)).transpose(1, 2)
    er, sr = ref(pts)
    ((er * ge).sum() + (sr * gs).sum()).backward()
    eo, so = my(pts)
    ((eo * ge).sum() + (so * gs).sum()).backward()
    close(eo, er, "dgcnn embedding", rtol=1e-5, atol=1e-6)
    close(so, sr, "dgcnn seg", rtol=1e-5, atol=1e-6)
    rg = {kk: p.grad for kk, p in ref.named_parameters()}
    names = sorted(rg)
    for kk, p in my.named_parameters():
        close(p.grad, rg[kk], "dgcnn d" + kk, rtol=1e-4, atol=1e-5 * max(v.abs().max().item() for v in rg.values()))
    save(name, seed=seed, knn_head=idx_r[:, :64].to(torch.int16), knn_sum=idx_r.sum(dim=(1, 2)),
         emb_head=er[:, :64].detach(), emb_sum=er.detach().sum(dim=1), seg=sr.detach(),
         grad_names=np.array(names), grad_norms=np.array([rg[n].norm().item() for n in names]),
         g_enc_conv1=rg["encoder.conv1.0.weight"], g_seg=rg["mlp_segmentation.weight"], g_emb=rg["mlp_seg_prob2.weight"])


def golden_dgcnn():
    """config 5: src/dgcnn.DGCNGn (k=20): outputs + gradients on B=2 x 1024 points, at the configuration's N = 2048,
    and the normals variant (num_channels=6: knn_points_normals, src/dgcnn.py:30-71,199-222) on B=2 x 512."""
    print("[dgcnn]")
    _golden_dgcnn_case("model_dgcnn", 2, 1024, 20, 41)
    _golden_dgcnn_case("model_dgcnn_2048", 2, 2048, 20, 43)
    _golden_dgcnn_case("model_dgcnn_normals", 2, 512, 20, 45, num_channels=6)


def golden_dgcnn_selfsup():
    """configs[4] end to end: reference DGCNGn (src/dgcnn.py:225-267) -> reference convex_loss (convex_loss.py:27-103),
    B = 2 x 2048 blob clouds, mean(total).backward().  The harness conventions of golden_selfsup_step: shared covariance
    noise, pinned SVD signs, the build's Fibonacci (U,V) table on both sides, the reference's representative ids
    (`center_ids`).  An untrained DGCNN embeds a shape as one cluster, so the bias-free embedding head `mlp_seg_prob2`
    (256 -> 128) is pre-conditioned in closed form (ridge regression, lambda = 1, fp64) from the reference's own features
    to one unit prototype per generating blob; the fitted head is part of the fixture."""
    import make_golden_fit as F_
    print("[dgcnn self-supervised]")
    CL = refshim.ref("convex_loss")
    EF = refshim.ref("src.ellipsoid_fitting")
    SE = refshim.ref("src.sample_ellipsoid")
    MS = refshim.ref("src.mean_shift")
    B, N, k, seed = 2, 2048, 20, 85      # (83 sat on a partition that flips with the CPU thread count: K 10 / 11 / 8; 84 and 85 give K = [8, 8] in fp32 at 1 and 8 threads and in fp64)
    D, ref_net, my_net = _dgcnn_pair(3, k, seed_w=37)
    cham_np, lab_np = synth.blobs_with_labels(B, 5000, seed)
    cham = torch.from_numpy(cham_np)
    sel = torch.from_numpy(np.random.default_rng(seed + 1).choice(5000, N, replace=False))
    xyz = cham[:, sel].transpose(1, 2).contiguous()
    cham_t = cham.transpose(1, 2).contiguous()
    R = torch.from_numpy(synth.uniform01((3, 3), seed))
    q, iters = 0.05, 10
    cap = {}
    h = ref_net.bn_seg_prob1.register_forward_hook(lambda m, i, o: cap.__setitem__("y", o))
    with torch.no_grad():
        ref_net(xyz)
    h.remove()
    A1 = torch.relu(cap["y"]).double().permute(0, 2, 1).reshape(-1, 256)
    proto = np.random.default_rng(seed + 5).normal(size=(8, 128))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    P = torch.from_numpy(proto)[torch.from_numpy(lab_np)[:, sel].reshape(-1)]
    sol = torch.linalg.solve(A1.T @ A1 + 1.0 * torch.eye(256, dtype=torch.float64), A1.T @ P)
    emb_W = sol.T.float().contiguous()                                     # [128, 256]
    with torch.no_grad():
        ref_net.mlp_seg_prob2.weight.copy_(emb_W.unsqueeze(-1))
        my_net.mlp_seg_prob2.weight.copy_(emb_W.unsqueeze(-1))

    def ref_customsvd_canonical(Mx):
        U, S, V = refshim.ref("src.fitting_utils").customsvd(Mx)
        sg = orc.canonical_signs(V).view(1, 3)
        return U * sg, S, V * sg

    def ref_sample(self, a, b_, c, center, transformation, n=500):
        U, V = orc.fibonacci_uv(int(n))
        p = self.uniform_sample_points_on_ellipsoid(U, V, a, b_, c)
        return p @ transformation.T + center, None

    real_nms = MS.MeanShift.nms
    ref_ids = []

    def recording_nms(self, centers, X, b):
        out3 = real_nms(self, centers, X, b)
        ref_ids.append(out3[1].clone())
        return out3

    ref_net.zero_grad()
    remb, rseg = ref_net(xyz)
    remb.retain_grad()
    with F_.patched(torch, "rand", lambda *a, **kw: R.clone()), F_.patched(MS.MeanShift, "nms", recording_nms), \
            F_.patched(EF, "customsvd", ref_customsvd_canonical), F_.patched(SE.SampleEllipsoid, "sample", ref_sample):
        rtot, rcham, rparams, rlabels = CL.convex_loss(xyz, cham_t, remb.permute(0, 2, 1), quantile=q, iterations=iters,
                                                       max_num_clusters=25)
    torch.mean(rtot).backward()
    assert len(ref_ids) == B, "a quantile-doubling retry happened: pick other inputs"
    rg = {kk: p.grad.detach().clone() for kk, p in ref_net.named_parameters() if p.grad is not None}

    my_net.zero_grad()
    oemb, oseg = my_net(xyz)
    oemb.retain_grad()
    otot, ocham, oparams, olabels = orc.convex_loss(xyz, cham_t, oemb.permute(0, 2, 1), quantile=q, iterations=iters,
                                                    max_num_clusters=25, rand_table=[[R] * 64] * B, canonical=True,
                                                    center_ids=ref_ids)
    torch.mean(otot).backward()
    og = {kk: p.grad.detach().clone() for kk, p in my_net.named_parameters() if p.grad is not None}
    close(oemb, remb, "dgcnn selfsup embedding", rtol=1e-4, atol=1e-5)
    Ks = [len(p) for p in rparams]
    assert Ks == [len(p) for p in oparams], (Ks, [len(p) for p in oparams])
    for b in range(B):
        assert F_.same_partition(rlabels[b], olabels[b]), "label partition differs b=%d" % b
    print("  ok partition      K per shape", Ks)
    assert min(Ks) >= 4, Ks
    close(otot, rtot, "dgcnn selfsup total_loss", rtol=1e-4)
    close(ocham, rcham, "dgcnn selfsup chamfer_loss", rtol=1e-4)
    print("  dX (embedding gradient) oracle-vs-reference rel L2 %.2e" % ((oemb.grad - remb.grad).norm() / remb.grad.norm()).item())
    names = sorted(rg)
    assert "mlp_segmentation.weight" not in names and "mlp_seg_prob2.weight" in names      # the seg head is off this path
    worst = 0.0
    for kk in names:
        rel = ((og[kk] - rg[kk]).norm() / max(rg[kk].norm().item(), 1e-30)).item()
        print("    grad %-32s |g| %.3e  oracle-vs-reference rel %.1e" % (kk, rg[kk].norm().item(), rel))
        worst = max(worst, rel)
        assert rel < 2e-2, (kk, rel)
    # Bars, MEASURED: the oracle once more in fp64 on the same inputs.  A DGCNN forward is sensitive to rounding in a way the
    # PointNet++ one is not -- its second kNN graph is built on computed features, a 1e-7 difference flips neighbours, the
    # embedding then moves by ~1e-3 (relative L2) and the gradients of the edge convolutions by a few per cent.  Also
    # measured (round 5, GPU box): the SAME fp32 oracle on another host (other thread count) gives a loss 3.1e-4 away.
    my64 = orc.OracleDGCNGn(128, 3, k).double()
    my64.load_state_dict({kk: v.double() for kk, v in my_net.state_dict().items()})
    e64, _ = my64(xyz.double())
    t64 = orc.convex_loss(xyz.double(), cham_t.double(), e64.permute(0, 2, 1), quantile=q, iterations=iters, max_num_clusters=25,
                          rand_table=[[R.double()] * 64] * B, canonical=True, center_ids=ref_ids)[0]
    torch.mean(t64).backward()
    g64 = {kk: p.grad.detach() for kk, p in my64.named_parameters() if p.grad is not None}
    dev_loss = abs(float(t64.detach()) - float(rtot.detach())) / abs(float(t64.detach()))
    dev_emb = float((e64.detach() - remb.detach().double()).norm() / e64.detach().norm())
    dev_grad = max(float((rg[kk].double() - g64[kk]).norm() / g64[kk].norm()) for kk in names)
    print("  fp32 reference vs fp64 oracle: loss %.2e, embedding %.2e (rel L2), worst parameter gradient %.2e" % (dev_loss, dev_emb, dev_grad))
    loss_bar = max(1e-3, 3.0 * dev_loss)      # floor: 3 x the host-to-host spread of the fp32 oracle itself (3.1e-4)
    grad_bar = max(2e-2, 2.0 * dev_grad, 2 * worst)
    save("step_dgcnn_selfsup", loss_bar=np.array(loss_bar), dev_fp64=np.array([dev_loss, dev_emb, dev_grad]), seed=seed, R=R, emb_W=emb_W, total_loss=rtot.detach(), chamfer_loss=rcham.detach(),
         K=np.array(Ks), labels=torch.stack(rlabels).to(torch.int16), emb_head=remb[:, :64].detach(),
         grad_names=np.array(names), grad_norms=np.array([rg[n].norm().item() for n in names]),
         g_emb_W=rg["mlp_seg_prob2.weight"], g_enc_conv1=rg["encoder.conv1.0.weight"],
         g_emb_head=remb.grad[:, :64].contiguous(), g_emb_norm=remb.grad.norm(), grad_bar=np.array(grad_bar),
         center_ids=torch.stack([torch.cat([i, torch.full((32 - i.shape[0],), -1, dtype=torch.long)]) for i in ref_ids]).to(torch.int16))


def golden_data():
    """pc_normalize (data_utils/ShapeNetDataLoader.py:17-22): reference vs oracle on a small cloud; then the reference's dataset
    classes, its two augmentation functions and its `evaluation` run on a synthetic tree (golden_data_readers / golden_eval)."""
    print("[data]")
    DL = refshim.ref("data_utils.ShapeNetDataLoader")
    pc = (synth.cloud("blobs", 1, 96, 5)[0] * 3.0 + np.array([[0.5, -1.0, 2.0]], dtype=np.float32)).astype(np.float32)
    ref = DL.pc_normalize(pc.copy())
    got = orc.pc_normalize_np(pc.copy())
    eq(torch.from_numpy(got), torch.from_numpy(ref), "pc_normalize")
    save("data_normalize", cloud=pc, normalized=ref)
    golden_data_readers(DL)
    golden_eval()


TREE_SEED, ACD_SEED, NPOINT = 21, 22, 48
ITEM_SEED = 1000          # np.random.seed(ITEM_SEED + index) in front of every __getitem__ (both sides)


def _rel(paths, root):
    return np.array([os.path.relpath(p, root) for p in paths])


def golden_data_readers(DL):
    """The reference's PartNormalDataset / SelfSupPartNormalDataset / ACDSelfSupDataset (ShapeNetDataLoader.py:24-140, :149-262,
    :265-412) on the seeded synthetic trees of prifit_amd/synth.py, every `__getitem__` under a replayed `np.random` state,
    and provider.random_scale_point_cloud / shift_point_cloud (provider.py:278-303) under a seeded state.  The fixture holds
    the reference's outputs; the trees are re-written from their seeds by the tests."""
    import random
    import tempfile
    provider = refshim.ref("provider")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root, acd = os.path.join(tmp, "shapenet"), os.path.join(tmp, "acd")
        synth.write_partseg_tree(root, TREE_SEED)
        for tag, kw in (("trainval", dict(split="trainval", normal_channel=False)), ("test_n", dict(split="test", normal_channel=True)),
                        ("train_car_chair", dict(split="train", normal_channel=False, class_choice=["Car", "Chair"]))):
            ds = DL.PartNormalDataset(root=root, npoints=NPOINT, **kw)
            out["pn_%s_paths" % tag] = _rel([fn for _, fn in ds.datapath], root)
            out["pn_%s_cats" % tag] = np.array([c for c, _ in ds.datapath])
            out["pn_%s_classes" % tag] = np.array(["%s=%d" % kv for kv in sorted(ds.classes.items())])
            for i in range(len(ds)):
                np.random.seed(ITEM_SEED + i)
                pts, cls, seg = ds[i]
                out["pn_%s_pts_%d" % (tag, i)], out["pn_%s_cls_%d" % (tag, i)], out["pn_%s_seg_%d" % (tag, i)] = pts, cls, seg
            out["pn_%s_n" % tag] = len(ds)
        # the few-shot draw (`random.sample`, :77-79) under a seeded `random`
        random.seed(5)
        ds = DL.PartNormalDataset(root=root, npoints=NPOINT, split="trainval", k_shot=1)
        out["pn_kshot_paths"] = _rel([fn for _, fn in ds.datapath], root)
        # the trainer's "dummy" self-supervision set: everything that is not labeled data (train:190-203)
        train = DL.PartNormalDataset(root=root, npoints=NPOINT, split="train", k_shot=-1)
        test = DL.PartNormalDataset(root=root, npoints=NPOINT, split="test")
        labeled = [f for v in test.meta.values() for f in v] + [f for v in train.meta.values() for f in v]
        ss = DL.SelfSupPartNormalDataset(root=root, npoints=NPOINT, split="trainval", labeled_fns=labeled)
        out["ss_paths"] = _rel([fn for _, fn in ss.datapath], root)
        for i in range(len(ss)):
            np.random.seed(ITEM_SEED + i)
            pts, cls, seg = ss[i]
            out["ss_pts_%d" % i], out["ss_cls_%d" % i], out["ss_seg_%d" % i] = pts, cls, seg
        out["ss_n"] = len(ss)
        # ACD self-supervision set with the trainer's overlap removal ('.txt' paths of labeled files against '.npy' tokens)
        overlap = ["chair_t0", "lamp_t2"]
        synth.write_acd_tree(acd, ACD_SEED, overlap_tokens=overlap)
        ad = DL.ACDSelfSupDataset(root=acd, npoints=NPOINT, exclude_fns=labeled, prefetch=False)
        out["acd_tokens"] = np.array([os.path.splitext(os.path.basename(fn))[0] for _, fn in ad.datapath])
        out["acd_cats"] = np.array([c for c, _ in ad.datapath])
        for i in range(len(ad)):
            tok = out["acd_tokens"][i]
            np.random.seed(ITEM_SEED + i)
            pts, cham, cls, seg = ad[i]
            out["acd_pts_" + tok], out["acd_cham_" + tok], out["acd_seg_" + tok] = pts, np.array(cham), seg
        out["acd_n"] = len(ad)
        assert not any(t in overlap for t in out["acd_tokens"]), "overlap removal"
    # augmentation: one seeded state, scale then shift as train:372-373 calls them
    batch = synth.cloud("blobs", 5, 40, 9)
    np.random.seed(77)
    aug = batch.copy()
    aug[:, :, 0:3] = provider.random_scale_point_cloud(aug[:, :, 0:3])
    aug[:, :, 0:3] = provider.shift_point_cloud(aug[:, :, 0:3])
    out["aug_in"], out["aug_out"] = batch, aug
    save("data_readers", **out)


def eval_args(**over):
    """The fields of args_parser.py's namespace that testing.evaluation reads (testing.py:55-139)."""
    import argparse
    a = dict(gpu=None, cudnn_off=False, eval_split="test", npoint=NPOINT, normal=False, batch_size=5, num_classes=16, num_parts=50,
             seed=3, pretrained_model=None, model="models.pointnet2_part_seg_msg", category=True, if_cuboid=False, quantile=0.05,
             msc_iterations=10, max_num_clusters=25, alpha=1.0, beta=1.0, embed=False, reconstruct=False, dgcnn_k=20)
    a.update(over)
    return argparse.Namespace(**a)


def golden_eval():
    """The reference's `testing.evaluation(args, epoch, classifier, metrics)` (testing.py:49-249) itself, run in a temporary
    working directory that holds the synthetic tree at the relative path it reads (:72), with a stub classifier
    (prifit_oracle.StubSegClassifier: logits = a fixed function of the points) and the resampling draws replayed from one
    seeded `np.random` state.  Two environment shims, both outside the metric arithmetic: `torch.utils.data.DataLoader` is
    forced to num_workers=0 (the draws then come from THIS process's np.random in dataset order) and `np.float` (removed from
    numpy 1.24 on, used at :230) is mapped to `float`."""
    import contextlib
    import io
    import re
    import tempfile
    T = refshim.ref("testing")
    real_loader = torch.utils.data.DataLoader

    def loader0(ds, **kw):
        kw["num_workers"] = 0
        return real_loader(ds, **kw)

    had_float = hasattr(np, "float")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "ShapeSelfSup/dataset/shapenetcore_partanno_segmentation_benchmark_v0_normal")
        synth.write_partseg_tree(root, TREE_SEED)
        os.chdir(tmp)
        torch.utils.data.DataLoader = loader0
        if not had_float:
            np.float = float
        try:
            net = orc.StubSegClassifier(50, seed=4)
            metrics = {"best_class_avg_miou": -1.0, "best_acc": 0.0, "best_epoch": 0, "best_instance_avg_miou": 0.0, "best_chamfer_loss": 1e9}
            np.random.seed(123)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ret = T.evaluation(eval_args(), 6, net, metrics)
            # a second call that must NOT update the running best (class-average mIoU below the stored best)
            keep = dict(ret, best_class_avg_miou=2.0)
            np.random.seed(123)
            with contextlib.redirect_stdout(io.StringIO()):
                ret2 = T.evaluation(eval_args(), 9, orc.StubSegClassifier(50, seed=4), dict(keep))
        finally:
            os.chdir(cwd)
            torch.utils.data.DataLoader = real_loader
            if not had_float:
                del np.float
    assert ret is metrics and ret2 == keep
    cat_iou = dict(re.findall(r"eval mIoU of (\w+)\s+([0-9.]+|nan)", buf.getvalue()))
    assert len(cat_iou) == 16, buf.getvalue()[-800:]
    names = sorted(cat_iou)
    print("  reference evaluation: acc %.6f  class mIoU %.6f  instance mIoU %.6f  loss %.6f  (epoch %d)" % (
        ret["best_acc"], ret["best_class_avg_miou"], ret["best_instance_avg_miou"], ret["best_chamfer_loss"], ret["best_epoch"]))
    # the forward keywords the reference passes (testing.py:139): names only
    kw_names = np.array(sorted(net.calls[0]))
    save("eval_metrics", accuracy=ret["best_acc"], class_avg_iou=ret["best_class_avg_miou"],
         instance_avg_iou=ret["best_instance_avg_miou"], chamfer_loss=ret["best_chamfer_loss"], best_epoch=ret["best_epoch"],
         category_names=np.array(names), category_iou=np.array([float(cat_iou[n]) for n in names]), forward_kwargs=kw_names,
         n_batches=len(net.calls))


if __name__ == "__main__":
    assert refshim.available(), "needs the reference tree"
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["index", "modules", "modules_real", "model", "step", "fit", "dgcnn", "data", "variants"]
    if "data" in which:
        golden_data()
    if "variants" in which:
        golden_variants()
    if "index" in which:
        golden_index_ops()
    if "modules" in which:
        golden_modules()
    if "modules_real" in which:
        golden_modules_real()
    if "model" in which:
        golden_model()
    if "step" in which:
        golden_selfsup_step()
    if "dgcnn" in which:
        golden_dgcnn()
    if "dgcnn_selfsup" in which:
        golden_dgcnn_selfsup()
    if "nms_pair" in which:
        import make_golden_fit
        make_golden_fit.run_nms_pair_and_epa_guard(save, eq, close)
    if "fit" in which:
        import make_golden_fit
        make_golden_fit.run(save, eq, close)
    if "ms_variants" in which or "fit" in which:
        import make_golden_fit
        make_golden_fit.run_mean_shift_variants(save, close)
    if "bw_over" in which or "fit" in which:
        import make_golden_fit
        make_golden_fit.run_bandwidth_over(save, close)
    if "center_grad" in which:
        import make_golden_fit
        make_golden_fit.run_center_grad(save, eq, close)
    if "many_clusters" in which:
        import make_golden_fit
        make_golden_fit.run_many_clusters(save, eq, close)
    if "prune" in which:
        import make_golden_fit
        make_golden_fit.run_prune(save, eq, close)
    if "intersections" in which:
        import make_golden_fit
        make_golden_fit.run_intersections(save, eq, close)
    print("all oracle-vs-reference checks passed; fixtures under", GOLD)
