"""Shared by the CPU (oracle) and GPU (HIP) tests of tests/golden/step_selfsup.npz: SURVEY 8a row a29 step (2), the
self-supervised training iteration of train_partseg_shapenet.py:436-451 through models/pointnet2_part_seg_msg.py:64-134,
captured from the reference by oracle/make_golden.py:golden_selfsup_step."""
import numpy as np
import torch

from prifit_amd import synth

B, N, Q, ITERS = 2, 2048, 0.05, 10


def _t(a):
    return torch.from_numpy(np.asarray(a))


def inputs(g):
    seed = int(g["seed"])
    cham = _t(synth.cloud("blobs", B, 5000, seed))
    sel = _t(np.random.default_rng(seed + 1).choice(5000, N, replace=False))
    xyz = cham[:, sel].transpose(1, 2).contiguous()
    return dict(xyz=xyz, cham=cham.transpose(1, 2).contiguous(), cls=torch.zeros(B, 1, 16), s1=_t(g["s1"]), s2=_t(g["s2"]),
                R=_t(g["R"]), center_ids=_t(g["center_ids"]).long())


def seeded_state(g, ctor):
    """The reference's seeded network (same construction order and seeds as the generator) with the fixture's
    pre-conditioned embedding head."""
    torch.manual_seed(24)
    net = ctor(50)
    synth.xavier_like_trainer(net)
    synth.perturb_bn(net, 9)
    with torch.no_grad():
        net.extra_conv_emb.weight.copy_(_t(g["emb_W"]).unsqueeze(-1))
        net.extra_conv_emb.bias.copy_(_t(g["emb_b"]))
    return net


def same_partition(la, lb):
    pairs = torch.unique(torch.stack([la.long(), lb.long()], 1), dim=0)
    return pairs.shape[0] == torch.unique(la).shape[0] == torch.unique(lb).shape[0]


def check(g, out, grads, params_before, params_after, beta, loss_tol=1e-4, grad_tol=2e-2):
    """out = the model's 8-tuple; grads / params_* = dicts name -> cpu tensor (None where no gradient)."""
    seg, _, feat, total, chamfer, labels, params, emb = out
    torch.testing.assert_close(total.reshape(-1).cpu(), _t(g["total_loss"]).reshape(-1), rtol=loss_tol, atol=1e-7)
    torch.testing.assert_close(chamfer.reshape(-1).cpu(), _t(g["chamfer_loss"]).reshape(-1), rtol=loss_tol, atol=1e-7)
    assert abs(beta - float(g["beta"])) < 1e-12                      # one `beta *= 0.99` per step (msg:96-99)
    torch.testing.assert_close(feat[:, :, :64].detach().cpu(), _t(g["feat_head"]), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(emb[:, :, :64].detach().cpu(), _t(g["emb_head"]), rtol=1e-3, atol=1e-4)
    ref_labels = _t(g["labels"]).long()
    K = [int(k) for k in g["K"]]
    assert [len(p) for p in params] == K
    for b in range(B):
        assert same_partition(labels[b].cpu(), ref_labels[b]), "label partition differs, shape %d" % b
        assert torch.equal(labels[b].cpu().long(), ref_labels[b])   # with the representatives pinned: the very labels
    # gradients (representative ids pinned, see the generator's docstring)
    names = [str(s) for s in g["grad_names"]]
    norms = dict(zip(names, g["grad_norms"]))
    assert grads["conv2.weight"] is None or float(grads["conv2.weight"].abs().max()) == 0.0     # seg head is off this path
    worst = 0.0
    for k in names:
        if k.endswith(".bias") and "conv" in k and k != "extra_conv_emb.bias":
            continue                                                   # in front of a train-mode BatchNorm: true value 0
        n = float(grads[k].norm())
        worst = max(worst, abs(n - norms[k]) / norms[k])
        assert abs(n - norms[k]) <= grad_tol * norms[k], (k, n, norms[k])
    for key, name in (("g_extra_conv_emb_weight", "extra_conv_emb.weight"), ("g_conv1_weight", "conv1.weight"),
                      ("g_sa1_first", "sa1.conv_blocks.0.0.weight")):
        ref = _t(g[key])
        rel = float((grads[name] - ref).norm() / ref.norm())
        worst = max(worst, rel)
        assert rel <= grad_tol, (name, rel)
    # parameter checksum after the Adam step (train:252-259).  Adam's first step moves every entry by
    # lr * g / (|g| + eps) ~ lr * sign(g): entries whose gradient is rounding noise move by +-lr at random, so the
    # checksum is the update's L2 norm per parameter (insensitive to those signs), and the embedding head is compared
    # entry by entry where its gradient is well above the noise
    upd = dict(zip([str(s) for s in g["upd_names"]], g["upd_norms"]))
    for k, u in upd.items():
        if k.endswith(".bias") and "conv" in k and k != "extra_conv_emb.bias":
            continue          # true gradient 0 (train-mode BatchNorm behind it): |g| ~ eps, the step size is noise too
        got = float((params_after[k] - params_before[k]).norm())
        assert abs(got - u) <= 0.05 * u + 1e-12, (k, got, u)
    gw = _t(g["g_extra_conv_emb_weight"])
    big = gw.abs() > 1e-2 * gw.abs().max()
    d = (params_after["extra_conv_emb.weight"] - _t(g["p_extra_conv_emb_weight"])).abs()[big]
    assert float(d.max()) < 2e-5, float(d.max())
    return worst
