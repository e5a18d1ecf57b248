"""Shared by the CPU (oracle) and GPU (HIP) tests of the DGCNN fixtures captured from the reference by
oracle/make_golden.py: golden_dgcnn (model_dgcnn*.npz: src/dgcnn.py:149-267 outputs and gradients, incl. the normals
variant :30-71,199-222) and golden_dgcnn_selfsup (step_dgcnn_selfsup.npz: configs[4] end to end, DGCNGn -> convex_loss.py:27-103)."""
import numpy as np
import torch

from prifit_amd import synth

Q, ITERS = 0.05, 10
CASES = {"model_dgcnn": (2, 1024, 20, 3), "model_dgcnn_2048": (2, 2048, 20, 3), "model_dgcnn_normals": (2, 512, 20, 6)}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def seeded_state(ctor, num_channels, k, seed_w=31):
    """The generator's seeded parameters (construction order and seeds of make_golden._dgcnn_pair)."""
    torch.manual_seed(seed_w)
    net = ctor(128, num_channels, k)
    for m in net.modules():
        if isinstance(m, torch.nn.GroupNorm):
            with torch.no_grad():
                gen = torch.Generator().manual_seed(m.num_channels)
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
    return net


def network_inputs(g, B, N, num_channels):
    seed = int(g["seed"])
    if num_channels == 6:
        pts = _t(synth.cloud("surface", B, N, seed))
        nrm = _t(synth.features(B, N, 3, seed + 7))
        nrm = nrm / nrm.norm(dim=2, keepdim=True)
        pts = torch.cat([pts, nrm], dim=2).transpose(1, 2).contiguous()
    else:
        pts = _t(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous()
    ge = _t(synth.features(B, N, 128, seed + 1))
    gs = _t(synth.features(B, N, 3, seed + 2)).transpose(1, 2)
    return pts, ge, gs


def check_network(g, emb, seg, grads, out_tol=1e-3, grad_tol=2e-2):
    """emb [B,N,128], seg [B,3,N] (cpu), grads name -> cpu tensor."""
    torch.testing.assert_close(emb[:, :64], _t(g["emb_head"]), rtol=out_tol, atol=out_tol)
    torch.testing.assert_close(emb.sum(dim=1), _t(g["emb_sum"]), rtol=out_tol, atol=50 * out_tol)
    torch.testing.assert_close(seg, _t(g["seg"]), rtol=out_tol, atol=out_tol)
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    for name, gr in grads.items():
        assert gr is not None and abs(gr.norm().item() - norms[name]) <= grad_tol * norms[name] + 1e-5, (name, gr.norm().item(), norms[name])
    for name, key in (("encoder.conv1.0.weight", "g_enc_conv1"), ("mlp_segmentation.weight", "g_seg"), ("mlp_seg_prob2.weight", "g_emb")):
        want = _t(g[key])
        assert (grads[name] - want).norm() <= grad_tol * want.norm(), name


def selfsup_inputs(g):
    B, N, seed = 2, 2048, int(g["seed"])
    cham = _t(synth.cloud("blobs", B, 5000, seed))
    sel = _t(np.random.default_rng(seed + 1).choice(5000, N, replace=False))
    xyz = cham[:, sel].transpose(1, 2).contiguous()
    ids = _t(g["center_ids"]).long()
    return dict(xyz=xyz, cham=cham.transpose(1, 2).contiguous(), R=_t(g["R"]), center_ids=[r[r >= 0] for r in ids])


def selfsup_state(g, ctor, k=20):
    net = seeded_state(ctor, 3, k, seed_w=37)
    with torch.no_grad():
        net.mlp_seg_prob2.weight.copy_(_t(g["emb_W"]).unsqueeze(-1))
    return net


def same_partition(la, lb):
    pairs = torch.unique(torch.stack([la.long(), lb.long()], 1), dim=0)
    return pairs.shape[0] == torch.unique(la).shape[0] == torch.unique(lb).shape[0]


def partition_agreement(la, lb):
    """Fraction of points on which two labelings agree once every cluster of `lb` is matched to the cluster of `la` it
    overlaps most (1.0 <=> the same partition)."""
    la, lb = la.long(), lb.long()
    conf = torch.zeros(int(lb.max()) + 1, int(la.max()) + 1, dtype=torch.long)
    conf.index_put_((lb, la), torch.ones_like(la), accumulate=True)
    return float(conf.max(dim=1)[0].sum()) / la.numel()


def check_selfsup(g, total, chamfer, params, labels, emb, grads, loss_tol=None, grad_tol=None, exact_labels=True):
    """total / chamfer [1,1], params list[B] of lists, labels list[B] of [N], emb [B,N,128] (cpu), grads name -> cpu tensor or None.
    Default bars: the fixture's, measured by the generator (fp32 reference against the fp64 oracle; a DGCNN forward re-builds
    its second kNN graph on computed features, so rounding flips neighbours and moves the embedding by ~1e-3).
    exact_labels=False (another arithmetic produced the embedding): the partitions agree on >= 99.5 % of the points."""
    grad_tol = float(g["grad_bar"]) if grad_tol is None else grad_tol
    loss_tol = float(g["loss_bar"]) if loss_tol is None else loss_tol
    torch.testing.assert_close(total.reshape(-1), _t(g["total_loss"]).reshape(-1), rtol=loss_tol, atol=1e-7)
    torch.testing.assert_close(chamfer.reshape(-1), _t(g["chamfer_loss"]).reshape(-1), rtol=loss_tol, atol=1e-7)
    # (relative L2: a neighbour flipping in the second kNN graph moves single entries by more than any elementwise bar;
    # the fp32 reference and the fp64 oracle are dev_fp64[1] = 3.5e-3 apart by this measure)
    eh = _t(g["emb_head"])
    assert float((emb[:, :64] - eh).norm() / eh.norm()) <= 2.0 * float(g["dev_fp64"][1])
    K = [int(k) for k in g["K"]]
    assert [len(p) for p in params] == K
    ref_labels = _t(g["labels"]).long()
    for b in range(len(K)):
        if exact_labels:
            assert same_partition(labels[b], ref_labels[b]), "label partition differs, shape %d" % b
            assert torch.equal(labels[b].long(), ref_labels[b])          # representatives pinned: the very labels
        else:
            assert partition_agreement(labels[b], ref_labels[b]) >= 0.995, (b, partition_agreement(labels[b], ref_labels[b]))
    names = [str(s) for s in g["grad_names"]]
    norms = dict(zip(names, g["grad_norms"]))
    for off in ("mlp_segmentation.weight", "mlp_segmentation.bias"):   # the seg head is off this path
        assert grads.get(off) is None or float(grads[off].abs().max()) == 0.0
    worst = 0.0
    for k in names:
        n = float(grads[k].norm())
        worst = max(worst, abs(n - norms[k]) / norms[k])
        assert abs(n - norms[k]) <= grad_tol * norms[k], (k, n, norms[k])
    for key, name in (("g_emb_W", "mlp_seg_prob2.weight"), ("g_enc_conv1", "encoder.conv1.0.weight")):
        ref = _t(g[key])
        rel = float((grads[name] - ref).norm() / ref.norm())
        worst = max(worst, rel)
        assert rel <= grad_tol, (name, rel)
    return worst
