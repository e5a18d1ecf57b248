"""GPU parity of the DGCNN backbone (BASELINE.json configs[4]; src/dgcnn.py) vs the oracle and the golden
vectors captured from the reference: kNN indices (bit-exact on xyz), edge features, conv+GroupNorm+
LeakyReLU+max blocks (forward and autograd) and the full DGCNGn network."""
import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def D(hiplib):
    assert torch.cuda.is_available()
    from prifit_amd.src import dgcnn
    return dgcnn


def test_knn_and_graph_feature(D, golden):
    g = golden("model_dgcnn")
    pts = _t(synth.cloud("surface", 2, 1024, int(g["seed"]))).transpose(1, 2).contiguous()
    idx = D.knn(pts.cuda(), 20, 20)
    assert torch.equal(idx.cpu()[:, :64], _t(g["knn_head"]).long()) and torch.equal(idx.cpu().sum(dim=(1, 2)), _t(g["knn_sum"]))
    assert torch.equal(idx.cpu(), orc.knn(pts, 20, 20))
    # dilation: every 2nd of the 40 nearest.  torch.topk leaves the order of exactly equal distances open, so
    # compare the selected DISTANCES (identical multiset <=> same neighbours up to exact ties)
    inner = -2 * torch.matmul(pts.transpose(2, 1), pts)
    xx = torch.sum(pts ** 2, dim=1, keepdim=True)
    pw = -xx - inner - xx.transpose(2, 1)
    got, want = D.knn(pts.cuda(), 20, 40).cpu(), orc.knn(pts, 20, 40)
    assert torch.equal(torch.gather(pw, 2, got), torch.gather(pw, 2, want))
    assert (got != want).float().mean() < 1e-3
    x = _t(synth.features(2, 1024, 64, 3)).transpose(1, 2).contiguous().requires_grad_(True)
    f_ref, i_ref = orc.graph_feature(x, 20, 20)
    xg = x.detach().cuda().requires_grad_(True)
    f, i = D.get_graph_feature(xg, 20, 20, idx=i_ref.cuda())
    torch.testing.assert_close(f.cpu(), f_ref, rtol=0, atol=0)
    w = _t(synth.features(1, 128 * 20, 1024 * 2, 4)).reshape(2, 128, 1024, 20)
    (f_ref * w).sum().backward()
    (f * w.cuda()).sum().backward()
    torch.testing.assert_close(xg.grad.cpu(), x.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("gn_kernels", [True, False])
@pytest.mark.parametrize("pool", [True, False])
def test_conv_groupnorm_block(D, pool, gn_kernels, monkeypatch):
    # gn_kernels: GroupNorm statistics -> coefficient tables by prifit_gn_finalize / _bwd_finalize (default) or the torch fp64 form
    monkeypatch.setattr(D, "_GN_KERNELS", gn_kernels)
    B, N, k, Cin, Cout, G = 2, 512, 4, 24, 32, 2
    rows = B * N * (k if pool else 1)
    x = _t(synth.features(1, rows, Cin, 5))[0]
    W = _t(synth.features(1, Cout, Cin, 6))[0] * 0.2
    bias = None if pool else _t(synth.features(1, 1, Cout, 7))[0, 0]
    gamma = _t(synth.features(1, 1, Cout, 8))[0, 0] * 0.5 + 0.8
    beta = _t(synth.features(1, 1, Cout, 9))[0, 0] * 0.2
    slope = 0.2 if pool else 0.0
    leaves = [t.clone().requires_grad_(True) for t in (x, W, gamma, beta)] + ([bias.clone().requires_grad_(True)] if bias is not None else [])
    xr, Wr, gr, br = leaves[:4]
    y = torch.nn.functional.linear(xr, Wr, leaves[4] if bias is not None else None)
    y = y.view(B, -1, Cout).permute(0, 2, 1)                      # [B, C, positions]
    y = torch.nn.functional.group_norm(y, G, gr, br, 1e-5)
    y = torch.nn.functional.leaky_relu(y, slope).permute(0, 2, 1)
    ref = y.reshape(B * N, k, Cout).max(dim=1)[0] if pool else y.reshape(rows, Cout)
    go = _t(synth.features(1, ref.shape[0], Cout, 10))[0]
    (ref * go).sum().backward()
    dl = [t.detach().cuda().requires_grad_(True) for t in leaves]
    cfg = {"groups": G, "rps": N * (k if pool else 1), "slope": slope, "pool_K": k if pool else 0, "eps": 1e-5}
    out = D.ConvGNActFn.apply(dl[0], dl[1], dl[4] if bias is not None else None, dl[2], dl[3], cfg)
    (out * go.cuda()).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    for a, b, name in zip(dl, leaves, ["x", "W", "gamma", "beta", "bias"]):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=2e-3, atol=2e-4 * b.grad.abs().max().item(), msg=lambda m, name=name: name + ": " + m)


def test_conv_groupnorm_block_with_per_sample_offset(D):
    """conv -> (+ one row per sample) -> GroupNorm -> ReLU with the offset folded into the coefficient tables
    (prifit_gn_finalize_offset; the decoder's first layer by linearity, src/dgcnn.py:253-257) against plain torch."""
    B, N, Cin, Cout, G = 3, 512, 24, 32, 4
    x = _t(synth.features(1, B * N, Cin, 15))[0]
    W = _t(synth.features(1, Cout, Cin, 16))[0] * 0.2
    off = _t(synth.features(1, B, Cout, 17))[0] * 1.5
    gamma = _t(synth.features(1, 1, Cout, 18))[0, 0] * 0.5 + 0.8
    beta = _t(synth.features(1, 1, Cout, 19))[0, 0] * 0.2
    leaves = [t.clone().requires_grad_(True) for t in (x, W, gamma, beta, off)]
    xr, Wr, gr, br, orf = leaves
    y = torch.nn.functional.linear(xr, Wr).view(B, N, Cout) + orf.view(B, 1, Cout)
    y = torch.nn.functional.group_norm(y.permute(0, 2, 1), G, gr, br, 1e-5)
    ref = torch.relu(y).permute(0, 2, 1).reshape(B * N, Cout)
    go = _t(synth.features(1, B * N, Cout, 20))[0]
    (ref * go).sum().backward()
    dl = [t.detach().cuda().requires_grad_(True) for t in leaves]
    cfg = {"groups": G, "rps": N, "slope": 0.0, "pool_K": 0, "eps": 1e-5}
    out = D.ConvGNActFn.apply(dl[0], dl[1], None, dl[2], dl[3], cfg, dl[4])
    (out * go.cuda()).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    for a, b, name in zip(dl, leaves, ["x", "W", "gamma", "beta", "offset"]):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=2e-3, atol=2e-4 * b.grad.abs().max().item(), msg=lambda m, name=name: name + ": " + m)


@pytest.mark.parametrize("case", ["bias", "offset", "bias_and_offset", "global_pool_bias"])
def test_bias_and_offset_gradients_come_from_the_statistics(D, case, monkeypatch):
    """d loss / d bias and d loss / d offset of a convolution in front of a GroupNorm are column sums of dY.  By default they
    come out of prifit_gn_bwd_finalize (sum_rows dY = ca sum Gm + cb sum Y + cd rows, the column sums of Y kept by the forward)
    with no pass over dY; PRIFIT_GN_COLSUMS=0 sums dY with torch.  Same numbers to fp32 rounding."""
    B, N, Cin, Cout, G = 4, 2048, 64, 256, 4
    pool = case == "global_pool_bias"
    x = _t(synth.features(1, B * N, Cin, 41))[0]
    W = _t(synth.features(1, Cout, Cin, 42))[0] * 0.2
    bias = _t(synth.features(1, 1, Cout, 43))[0, 0] if "bias" in case else None
    off = _t(synth.features(1, B, Cout, 44))[0] * 1.5 if "offset" in case else None
    gamma = _t(synth.features(1, 1, Cout, 45))[0, 0] * 0.5 + 0.8
    beta = _t(synth.features(1, 1, Cout, 46))[0, 0] * 0.2
    go = _t(synth.features(1, B if pool else B * N, Cout, 47))[0].cuda()
    cfg = {"groups": G, "rps": N, "slope": 0.0, "pool_K": N if pool else 0, "eps": 1e-5}
    grads = []
    for colsums in (True, False):
        monkeypatch.setattr(D, "_GN_COLSUMS", colsums)
        dl = [None if t is None else t.detach().cuda().requires_grad_(True) for t in (x, W, bias, gamma, beta, off)]
        out = D.ConvGNActFn.apply(dl[0], dl[1], dl[2], dl[3], dl[4], cfg, dl[5])
        (out * go).sum().backward()
        grads.append([None if t is None else t.grad.clone() for t in dl])
    for a, b, name in zip(grads[0], grads[1], ["x", "W", "bias", "gamma", "beta", "offset"]):
        if a is None:
            assert b is None
            continue
        if name in ("bias", "offset"):
            assert a.abs().max().item() > 0
            torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * b.abs().max().item(), msg=lambda m, name=name: name + ": " + m)
        else:            # (the other gradients do not depend on the switch; the weight gradient's split-K sums are not bit-stable)
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6 * b.abs().max().item(), msg=lambda m, name=name: name + ": " + m)


@pytest.mark.parametrize("Cin,alg,nostore", [(64, True, True), (256, True, True), (256, True, False), (64, False, True)])
def test_global_max_pool_fused_into_the_block(D, Cin, alg, nostore, monkeypatch):
    """relu(gn(conv(.))) followed by the max over the whole cloud (src/dgcnn.py:194-197) as one pooled block with K = N
    (candidates from the product's epilogue, gradient routed through the winners) against plain torch.  alg: the backward in
    the algebraic form (default: per-sample [Cin, Cin] products + B x Cout winners' rows, no [B N, Cout] tensor dY) or through
    pool_bwd_apply and the two dense products over dY."""
    monkeypatch.setattr(D, "_GLOBAL_POOL_ALG", alg)
    monkeypatch.setattr(D, "_GLOBAL_POOL_NOSTORE", nostore)     # (with the algebraic backward: the product is not stored at all)
    B, N, Cout, G = 6, 2048, 1024, 8
    assert D.pool_product_ok(B * N, Cout, Cin)
    x = _t(synth.features(1, B * N, Cin, 25))[0]
    W = _t(synth.features(1, Cout, Cin, 26))[0] * 0.2
    bias = _t(synth.features(1, 1, Cout, 27))[0, 0]
    gamma = _t(synth.features(1, 1, Cout, 28))[0, 0] * 0.5 + 0.8
    beta = _t(synth.features(1, 1, Cout, 29))[0, 0] * 0.2
    leaves = [t.clone().requires_grad_(True) for t in (x, W, gamma, beta, bias)]
    xr, Wr, gr, br, bi = leaves
    y = torch.nn.functional.linear(xr, Wr, bi).view(B, N, Cout).permute(0, 2, 1)
    y = torch.relu(torch.nn.functional.group_norm(y, G, gr, br, 1e-5))
    ref = y.max(dim=2)[0]                                                  # [B, Cout]
    go = _t(synth.features(1, B, Cout, 30))[0]
    (ref * go).sum().backward()
    dl = [t.detach().cuda().requires_grad_(True) for t in leaves]
    cfg = {"groups": G, "rps": N, "slope": 0.0, "pool_K": N, "eps": 1e-5}
    out = D.ConvGNActFn.apply(dl[0], dl[1], dl[4], dl[2], dl[3], cfg)
    assert out.shape == (B, Cout)
    (out * go.cuda()).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    for a, b, name in zip(dl, leaves, ["x", "W", "gamma", "beta", "bias"]):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=2e-3, atol=2e-4 * b.grad.abs().max().item(), msg=lambda m, name=name: name + ": " + m)


def test_edge_conv_by_linearity_equals_rows_and_product(D, monkeypatch):
    """The edge convolution W [x_j - x_i | x_i] = U_j - Vc_i by per-point tables (csrc/edge_conv.hip, the default: no per-edge
    tensor in either direction) against the stored pre-activations + fused scatter (PRIFIT_EDGE_TABLES=0), the unfused
    backward, and the materialised edge rows + product over B N k rows (PRIFIT_EDGE_LINEARITY=0): same outputs and gradients
    to rounding."""
    B, N, k = 2, 512, 20
    torch.manual_seed(7)
    enc = D.DGCNNEncoderGn(input_channels=3, nn_nb=k).cuda()
    pts = _t(synth.cloud("surface", B, N, 41)).cuda()
    feats = _t(synth.features(B, N, 64, 42)).cuda()
    res = {}
    for arm in ("tables", True, "unfused", False):
        monkeypatch.setattr(D, "_EDGE_LINEARITY", bool(arm))
        monkeypatch.setattr(D, "_EDGE_FUSED_BWD", arm in (True, "tables"))   # "unfused": apply pass + scatter as two launches
        enc.zero_grad()
        with torch.no_grad():
            idx = D._knn_cl(pts, k)
            csr = D.edge_csr(idx) if arm == "tables" else None     # the default: per-point tables, no per-edge tensor
        f = feats.clone().requires_grad_(True)
        x1 = enc._edge_conv(pts, idx, enc.conv1, N, csr)
        x2 = enc._edge_conv(f, idx, enc.conv2, N, csr)
        go1 = _t(synth.features(1, B * N, 64, 43))[0].cuda()
        ((x1 + x2) * go1).sum().backward()
        res[arm] = (x1.detach(), x2.detach(), f.grad.clone(), enc.conv1[0].weight.grad.clone(), enc.conv2[0].weight.grad.clone(),
                    enc.bn2.weight.grad.clone())
    for other in (True, "unfused", False):
        for a, b, name in zip(res["tables"], res[other], ["x1", "x2", "dfeat", "dW1", "dW2", "dgamma2"]):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4 * max(1.0, b.abs().max().item()),
                                       msg=lambda m, name=name: "%s vs %s: %s" % (name, other, m))


def test_edge_csr_lists(D):
    """prifit_edge_csr: the in-edge lists (edge numbers i k + j) of a neighbour graph (as sets; out-of-range entries dropped)."""
    B, N, k = 3, 256, 7
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, N, (B, N, k), generator=gen, dtype=torch.int32)
    idx[0, 5, 2] = -1
    idx[1, 9, 0] = N
    idx[2, :, 0] = 17                                         # a hub: every centre of shape 2 points at 17
    offs, lst, pos = (t.cpu() for t in D.edge_csr(idx.cuda()))
    for b in range(B):
        assert offs[b, 0] == 0 and (offs[b, 1:] >= offs[b, :-1]).all()
        valid = (idx[b] >= 0) & (idx[b] < N)
        assert int(offs[b, N]) == int(valid.sum())
        edge = torch.arange(N * k).view(N, k)
        for n in (0, 17, 100, N - 1):
            want = sorted(edge[valid & (idx[b] == n)].tolist())
            assert sorted(lst[b, offs[b, n]:offs[b, n + 1]].tolist()) == want
        flat_ok = valid.view(-1)
        assert (pos[b][~flat_ok] == -1).all()
        assert torch.equal(lst[b][pos[b][flat_ok].long()], torch.arange(N * k, dtype=torch.int32)[flat_ok])


@pytest.mark.parametrize("stacked", [False, True])
@pytest.mark.parametrize("C", [64, 128, 256])
def test_edge_tables_forward_and_backward_against_torch(D, C, stacked):
    """EdgeConvTabFn (tables + CSR gather) against the same block written in torch on the explicit [B, N, k, C] tensor,
    GroupNorm with negative and zero-crossing scales, an out-of-range neighbour, ties between neighbours.  stacked: the
    operands are the halves of one [B, N, 2C] tensor [U | Vb], y = U_j - U_i + Vb_i, one gradient tensor back."""
    B, N, k, G = 2, 128, 9, 2
    gen = torch.Generator().manual_seed(C)
    U = torch.randn(B, N, C, generator=gen)
    V2 = torch.randn(B, N, C, generator=gen) * 0.5          # the centre term Vc, or Vb when stacked
    idx = torch.randint(0, N, (B, N, k), generator=gen, dtype=torch.int32)
    idx[0, 3, 4] = -1                                        # zero row, no gradient
    idx[1, 7, 5] = idx[1, 7, 1]                              # a repeated neighbour: tie, the first position wins
    gamma = torch.randn(C, generator=gen)                    # both signs
    beta = torch.randn(C, generator=gen) * 0.3
    go = torch.randn(B * N, C, generator=gen)
    cfg = {"groups": G, "rps": N * k, "slope": 0.2, "pool_K": k, "eps": 1e-5}
    first = torch.cat([U, V2], dim=2) if stacked else U
    leaves = [t.clone().requires_grad_(True) for t in (first, V2, gamma, beta)]
    Fr, Vr, gr, br = leaves
    Ur = Fr[:, :, :C]
    centre = Ur - Fr[:, :, C:] if stacked else Vr
    ok = ((idx >= 0) & (idx < N)).unsqueeze(-1)
    rows = torch.gather(Ur.unsqueeze(1).expand(B, N, N, C), 2, idx.clamp(0, N - 1).long().unsqueeze(-1).expand(B, N, k, C))
    y = torch.where(ok, rows - centre.unsqueeze(2), torch.zeros(()))
    yn = torch.nn.functional.group_norm(y.permute(0, 3, 1, 2), G, gr, br, 1e-5)
    ref = torch.nn.functional.leaky_relu(yn, 0.2).max(dim=3)[0].permute(0, 2, 1).reshape(B * N, C)
    (ref * go).sum().backward()
    dl = [t.detach().clone().cuda().requires_grad_(True) for t in (first, V2, gamma, beta)]
    idx_d = idx.cuda()
    out = D.EdgeConvTabFn.apply(dl[0], None if stacked else dl[1], idx_d, D.edge_csr(idx_d), dl[2], dl[3], cfg)
    (out * go.cuda()).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    for a, b, name in zip(dl, leaves, ["U or [U|Vb]", "Vc", "gamma", "beta"]):
        if stacked and name == "Vc":
            continue
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-3, atol=1e-4 * b.grad.abs().max().item(),
                                   msg=lambda m, name=name: name + ": " + m)


@pytest.mark.parametrize("name", ["model_dgcnn", "model_dgcnn_2048", "model_dgcnn_normals"])
def test_dgcnn_network(D, golden, name):
    """DGCNGn (src/dgcnn.py:225-267) against the reference's outputs and gradients: B = 2 x 1024, the configuration's own
    N = 2048, and the normals variant (num_channels = 6: knn_points_normals :30-71, encoder branch :199-222)."""
    import dgcnn_common as C
    g = golden(name)
    B, N, k, ch = C.CASES[name]
    ref = C.seeded_state(orc.OracleDGCNGn, ch, k)
    net = D.DGCNGn(emb_size=128, num_channels=ch, nn_nb=k)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    if ch == 3:
        assert sum(p.numel() for p in net.parameters()) == 1179267  # DGCNN model size
    net.load_state_dict(ref.state_dict())
    net.cuda()
    pts, ge, gs = C.network_inputs(g, B, N, ch)
    if ch == 6:    # the first graph under the normal-weighted metric: the reference's very neighbours
        idx = D.knn_points_normals(pts.cuda(), k, k).cpu()
        assert torch.equal(idx, orc.knn_points_normals(pts, k, k))
        assert torch.equal(idx[:, :64], _t(g["knn_head"]).long()) and torch.equal(idx.sum(dim=(1, 2)), _t(g["knn_sum"]))
        f = D.get_graph_feature_with_normals(pts.cuda(), k, k)
        assert f.shape == (B, 12, N, k)
        torch.testing.assert_close(f.cpu(), orc.graph_feature(pts, k, k, normals=True)[0], rtol=0, atol=0)
    emb, seg = net(pts.cuda())
    assert emb.shape == (B, N, 128) and seg.shape == (B, 3, N)
    ((emb * ge.cuda()).sum() + (seg * gs.cuda()).sum()).backward()
    C.check_network(g, emb.detach().cpu(), seg.detach().cpu(), {n_: p.grad.cpu() for n_, p in net.named_parameters()},
                    out_tol=1e-3, grad_tol=2e-2)


def test_dgcnn_with_convex_loss_config5(D, golden):
    """configs[4] end to end against the reference (tests/golden/step_dgcnn_selfsup.npz: reference DGCNGn -> reference
    convex_loss, B = 2 x 2048, q = 0.05, 10 mean-shift iterations): loss, K, the very labels, every parameter-gradient norm
    and the embedding head's / first edge convolution's gradient vectors -- through the adapter's own forward."""
    import dgcnn_common as C
    g = golden("step_dgcnn_selfsup")
    d = C.selfsup_inputs(g)
    ref = C.selfsup_state(g, orc.OracleDGCNGn)
    net = D.get_model(50, k=20)
    net.net.load_state_dict(ref.state_dict())
    net.cuda()
    out = net(d["xyz"].cuda(), None, chamfer_points=d["cham"].cuda(), include_convex_loss=True, quantile=C.Q,
              msc_iterations=C.ITERS, max_num_clusters=25,
              fit_inputs=dict(rand_table=d["R"].cuda(), canonical=True, center_ids=torch.from_numpy(np.asarray(g["center_ids"])).long()))
    B, N = 2, 2048
    assert len(out) == 8 and out[0].shape == (B, N, 3) and out[2].shape == (B, 128, N)
    total, chamfer, labels, params, fe = out[3], out[4], out[5], out[6], out[7]
    total.mean().backward()
    grads = {k_: (None if p.grad is None else p.grad.detach().cpu()) for k_, p in net.net.named_parameters()}
    worst = C.check_selfsup(g, total.detach().cpu(), chamfer.detach().cpu(), params, [l.cpu() for l in labels],
                            fe.detach().permute(0, 2, 1).cpu(), grads, exact_labels=False)
    print("configs[4] step vs reference: worst relative gradient deviation %.2e (bar %.2e), loss bar %.1e"
          % (worst, float(g["grad_bar"]), float(g["loss_bar"])))
    assert abs(net.beta - 0.99) < 1e-12
    # The loss bar above is the backbone's (rounding flips kNN neighbours: generator docstring).  The FIT path itself is held
    # to the north star's 1e-4: both sides on the SAME embedding -- the oracle's DGCNN run on this box's CPU
    from prifit_amd.convex_loss import convex_loss
    with torch.no_grad():
        emb_o, _ = ref(d["xyz"])
    kw = dict(quantile=C.Q, iterations=C.ITERS, max_num_clusters=25, canonical=True)
    Xo = emb_o.permute(0, 2, 1).contiguous().requires_grad_(True)
    to, _, po, lo = orc.convex_loss(d["xyz"], d["cham"], Xo, rand_table=[[d["R"]] * 64] * 2, center_ids=d["center_ids"], **kw)
    to.mean().backward()
    Xh = emb_o.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    th, _, ph, lh = convex_loss(d["xyz"].cuda(), d["cham"].cuda(), Xh, rand_table=d["R"].cuda(),
                                center_ids=torch.from_numpy(np.asarray(g["center_ids"])).long(), **kw)
    th.mean().backward()
    assert [len(p) for p in ph] == [len(p) for p in po]
    assert abs(float(th.detach()) - float(to.detach())) <= 1e-4 * abs(float(to.detach())), (float(th.detach()), float(to.detach()))
    assert all(torch.equal(a.cpu().long(), b_) for a, b_ in zip(lh, lo))
    assert (Xh.grad.cpu() - Xo.grad).norm() <= 2e-2 * Xo.grad.norm()


@pytest.mark.parametrize("N,k", [(2048, 20), (1024, 40), (300, 7), (64, 64), (70, 3), (2048, 40)])
def test_knn3_from_the_cloud_equals_product_plus_selection(D, N, k, monkeypatch):
    """prifit_knn3_topk (the first graph straight from the xyz cloud, no pairwise matrix) against the product kernel +
    prifit_knn_topk it replaces: the same indices bit for bit -- on surface clouds, on clouds with every point twice and a
    ten-fold point (exact ties across the k-th place), ragged N."""
    B = 3
    rng = np.random.default_rng(7 * N + k)
    xs = [synth.cloud("surface", B, N, N + k), rng.normal(size=(B, N, 3)).astype(np.float32)]
    xs[1][:, N // 2:] = xs[1][:, :N - N // 2]
    xs[1][0, :10] = xs[1][0, 0]
    for x in xs:
        xt = torch.from_numpy(x).cuda()
        monkeypatch.setattr(D, "_KNN3_FUSED", True)
        got = D._knn_cl(xt, k)
        monkeypatch.setattr(D, "_KNN3_FUSED", False)
        want = D._knn_cl(xt, k)
        assert torch.equal(got, want)


@pytest.mark.parametrize("N,C,k", [(2048, 64, 20), (1024, 64, 40), (256, 128, 16)])
def test_feature_graph_on_the_symmetric_product_kernel(D, N, C, k, monkeypatch):
    """The pairwise matrix of a neighbour graph over FEATURES (the second graph of the encoder, src/dgcnn.py:189-191, C = 64) on
    the symmetric kernel (prifit_gram_sym_f32: tiles on and above the diagonal, the others as transposes) against the general
    product: the same matrix and the same neighbour lists bit for bit, duplicated points included."""
    B = 3
    x = torch.from_numpy(np.random.default_rng(N + C).normal(size=(B, N, C)).astype(np.float32)).cuda()
    x[:, N // 2:N // 2 + 8] = x[:, :8]
    import ctypes
    from prifit_amd._lib import call, cur_stream, ptr
    from prifit_amd.nn_ops import NT, gemm
    LL = ctypes.c_longlong
    G0, G1 = (torch.empty(B, N, N, device="cuda") for _ in range(2))
    gemm(NT, N, N, C, x, C, x, C, G0, N, batch=B, sA=N * C, sB=N * C, sC=N * N)
    call("prifit_gram_sym_f32", ptr(x), LL(C), LL(N * C), ptr(G1), LL(N), LL(N * N), N, C, B, cur_stream())
    assert torch.equal(G0, G1)
    monkeypatch.setattr(D, "_KNN_GRAM_SYM", True)
    got = D._knn_cl(x, k)
    monkeypatch.setattr(D, "_KNN_GRAM_SYM", False)
    assert torch.equal(got, D._knn_cl(x, k))


@pytest.mark.parametrize("N,k", [(2048, 20), (1024, 40), (300, 7), (64, 64), (70, 3)])
def test_knn_selection_order_and_ties(hiplib, N, k):
    """prifit_knn_topk (selection + one sort per wave) against a stable sort of the same values: descending value, exact ties
    to the lower index -- on clouds with duplicated points (many exact ties, also across the k-th place), ragged N, k = N."""
    import ctypes
    from prifit_amd._lib import call, cur_stream, ptr
    B = 3
    rng = np.random.default_rng(N + k)
    x = rng.normal(size=(B, N, 3)).astype(np.float32)
    x[:, N // 2:] = x[:, :N - N // 2]                      # every point twice: equal distances everywhere
    x[0, :10] = x[0, 0]                                    # and a ten-fold point
    xt = torch.from_numpy(x).cuda()
    G = torch.matmul(xt, xt.transpose(1, 2)).contiguous()  # [B, N, N] inner products
    xx = (xt * xt).sum(-1).contiguous()
    idx = torch.empty(B, N, k, dtype=torch.int32, device="cuda")
    call("prifit_knn_topk", ptr(G), ptr(xx), B, N, k, ptr(idx), cur_stream())
    v = ((-xx).unsqueeze(2) - (-2.0 * G)) - xx.unsqueeze(1)                       # the kernel's expression, same rounding
    want = torch.argsort(-v.cpu(), dim=2, stable=True)[:, :, :k]
    assert torch.equal(idx.cpu().long(), want)
