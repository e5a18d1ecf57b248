"""CPU: the oracle (C + torch restatements) against the golden vectors captured from the reference."""
import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("kind,N", [("cube", 2048), ("surface", 2048), ("cube", 1024), ("surface", 1024)])
def test_index_ops_match_reference(golden, kind, N):
    g = golden(f"index_{kind}_n{N}")
    seed = int(g["seed"])
    B = 4
    xyz = _t(synth.cloud(kind, B, N, seed))
    f1 = orc.c_farthest_point_sample(xyz, 512, _t(g["start1"]))
    assert torch.equal(f1, _t(g["fps1"]).long())
    assert torch.equal(orc.farthest_point_sample(xyz[:2], 512, _t(g["start1"])[:2]), _t(g["fps1"]).long()[:2])
    c1 = orc.gather_rows(xyz, f1)
    f2 = orc.c_farthest_point_sample(c1, 128, _t(g["start2"]))
    assert torch.equal(f2, _t(g["fps2"]).long())
    c2 = orc.gather_rows(c1, f2)
    for lname, pts, ctr, rs in (("sa1", xyz, c1, [(0.1, 32), (0.2, 64), (0.4, 128), (0.2, 32)]),
                                ("sa2", c1, c2, [(0.4, 64), (0.8, 128)])):
        for r, k in rs:
            for fn in (orc.c_query_ball_point, orc.query_ball_point):
                gi = fn(r, k, pts, ctr)
                assert torch.equal(gi[:, :64], _t(g[f"ball_{lname}_{r}_{k}_head"]).long())
                assert torch.equal(gi.sum(dim=(1, 2)), _t(g[f"ball_{lname}_{r}_{k}_sum"]))
                assert torch.equal((gi * (torch.arange(k) + 1)).sum(dim=(1, 2)), _t(g[f"ball_{lname}_{r}_{k}_wsum"]))
    for lname, a, b in (("fp1", xyz, c1), ("fp2", c1, c2)):
        d3, i3 = orc.c_three_nn(a, b)
        assert torch.equal(i3, _t(g[f"nn3_{lname}_idx"]).long())
        assert torch.equal(d3, _t(g[f"nn3_{lname}_d"]))
        d3t, i3t = orc.three_nn(a, b)
        assert torch.equal(i3t, i3) and torch.equal(d3t, d3)
    assert torch.equal(orc.c_square_distance(c1, c2)[:2, :128], _t(g["sqdist_fp2_head"]))


def test_ball_query_edge_cases():
    # empty ball (query far away) keeps N everywhere, like the reference's sort-based code
    xyz = torch.zeros(1, 8, 3)
    q = torch.full((1, 2, 3), 5.0)
    out = orc.c_query_ball_point(0.1, 4, xyz, q)
    assert torch.equal(out, torch.full((1, 2, 4), 8, dtype=torch.int64))
    assert torch.equal(orc.query_ball_point(0.1, 4, xyz, q), out)
    # more in-ball points than nsample: ordered truncation
    q0 = torch.zeros(1, 1, 3)
    assert orc.c_query_ball_point(0.1, 4, xyz, q0).tolist() == [[[0, 1, 2, 3]]]


def _load_state(mod, g, prefix="g_"):
    return {k[len(prefix):]: _t(g[k]) for k in g.files if k.startswith(prefix)}


def test_modules_match_reference(golden):
    B, N = 2, 512
    g = golden("module_sa_msg")
    seed = int(g["seed"])
    xyz = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    feat = _t(synth.features(B, N, 16, seed)).transpose(1, 2).contiguous().requires_grad_(True)
    torch.manual_seed(11)
    sa = orc.OracleSetAbstractionMsg(64, [0.2, 0.4], [8, 16], 16, [[16, 32], [16, 24, 32]])
    synth.perturb_bn(sa, 3)
    nx, out = sa(xyz, feat, _t(g["start"]))
    gout = _t(synth.features(B, 64, out.shape[1], seed + 1)).transpose(1, 2)
    (out * gout).sum().backward()
    assert torch.equal(nx, _t(g["new_xyz"]))
    torch.testing.assert_close(out, _t(g["out"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(feat.grad, _t(g["dfeat"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sa.bn_blocks[0][0].running_mean, _t(g["running_mean_00"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sa.bn_blocks[0][0].running_var, _t(g["running_var_00"]), rtol=1e-5, atol=1e-6)
    grads = _load_state(sa, g)
    gmax = max(v.abs().max().item() for v in grads.values())
    for k, p in sa.named_parameters():
        torch.testing.assert_close(p.grad, grads[k], rtol=2e-4, atol=2e-5 * gmax)


def test_model_matches_reference(golden):
    g = golden("model_msg_sup")
    B, N, seed = 2, 2048, int(g["seed"])
    torch.manual_seed(21)
    net = orc.OracleMSGPartSeg(50)
    synth.xavier_like_trainer(net)
    synth.perturb_bn(net, 8)
    net.train()
    net.drop1.eval()
    xyz = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    cls = torch.zeros(B, 1, 16)
    cls[:, 0, 3] = 1.0
    target = _t(synth.labels(B, N, 50, seed))
    seg, (l1, l2, l3), feat, _, _ = net(xyz, cls, fps_start=(_t(g["s1"]), _t(g["s2"])))
    loss = orc.seg_loss(seg.reshape(-1, 50), target.view(-1))
    loss.backward()
    torch.testing.assert_close(loss.detach(), _t(g["loss"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(seg[:, :64].detach(), _t(g["seg_head"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(feat[:, :, :64].detach(), _t(g["feat_head"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(l3.detach(), _t(g["l3"]), rtol=1e-4, atol=1e-4)
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    for k, p in net.named_parameters():
        if k in norms and not (k.endswith(".bias") and "conv" in k and k != "conv2.bias") and norms[k] > 0:
            assert abs(p.grad.norm().item() - norms[k]) <= 3e-2 * norms[k], k
    torch.testing.assert_close(net.conv2.weight.grad, _t(g["g_conv2_weight"]), rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("i", [0, 1])
def test_real_shape_sa_levels_match_reference(golden, i):
    """SA1 / SA2 of the MSG network at their real shapes (B=4): the oracle against the reference's outputs and
    gradients (tests/golden/module_sa_real.npz)."""
    import sa_real_common as C
    g = golden("module_sa_real")
    xyz, feat, start, gout = C.inputs(g, i)
    sa = C.seeded_module(orc.OracleSetAbstractionMsg, i)
    f = feat.clone().requires_grad_(True)
    nx, out = sa(xyz, f, start)
    (out * gout).sum().backward()
    last = sa.bn_blocks[-1][-1]
    C.check(g, i, nx, out.detach(), f.grad, {k: p.grad for k, p in sa.named_parameters()},
            (last.running_mean, last.running_var), out_tol=2e-5)


def test_selfsup_training_step_matches_reference(golden):
    """zero_grad, forward with the convex loss on a 2048-subset of the chamfer points,
    mean(loss) * lambda, backward, Adam step (train_partseg_shapenet.py:436-451) -- the oracle against the values
    captured from the reference's own network file (tests/golden/step_selfsup.npz)."""
    import step_selfsup_common as C
    g = golden("step_selfsup")
    d = C.inputs(g)
    net = C.seeded_state(g, orc.OracleMSGPartSeg)
    net.train()
    net.drop1.eval()
    opt = torch.optim.Adam(net.parameters(), lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    opt.zero_grad()
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    out = net(d["xyz"], d["cls"], chamfer_points=d["cham"], include_convex_loss=True, quantile=C.Q, msc_iterations=C.ITERS,
              max_num_clusters=25, fps_start=(d["s1"], d["s2"]),
              fit_inputs=dict(rand_table=[[d["R"]] * 64] * C.B, canonical=True,
                              center_ids=[r[r >= 0] for r in d["center_ids"]]))
    (torch.mean(out[3]) * 1.0).backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in net.named_parameters()}
    opt.step()
    after = {k: p.detach().clone() for k, p in net.named_parameters()}
    C.check(g, out, grads, before, after, net.beta, loss_tol=1e-5, grad_tol=5e-3)
