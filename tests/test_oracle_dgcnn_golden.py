"""CPU: the oracle's DGCNN (oracle/prifit_oracle.py OracleDGCNGn, src/dgcnn.py:9-267) and the configs[4] pipeline
DGCNGn -> convex_loss against the values captured from the reference (tests/golden/model_dgcnn*.npz,
step_dgcnn_selfsup.npz; generator oracle/make_golden.py golden_dgcnn / golden_dgcnn_selfsup)."""
import pytest
import torch

import dgcnn_common as C
import prifit_oracle as orc


@pytest.mark.parametrize("name", sorted(C.CASES))
def test_dgcnn_network_matches_reference(golden, name):
    g = golden(name)
    B, N, k, ch = C.CASES[name]
    net = C.seeded_state(orc.OracleDGCNGn, ch, k)
    pts, ge, gs = C.network_inputs(g, B, N, ch)
    idx = orc.knn_points_normals(pts, k, k) if ch == 6 else orc.knn(pts, k, k)
    assert torch.equal(idx[:, :64], C._t(g["knn_head"]).long()) and torch.equal(idx.sum(dim=(1, 2)), C._t(g["knn_sum"]))
    emb, seg = net(pts)
    ((emb * ge).sum() + (seg * gs).sum()).backward()
    C.check_network(g, emb.detach(), seg.detach(), {k_: p.grad for k_, p in net.named_parameters()}, out_tol=1e-4, grad_tol=5e-3)   # (CPU sums are thread-order dependent: a max-pool winner flipping moves 1e-3 of a gradient)


def test_dgcnn_selfsup_step_matches_reference(golden):
    """configs[4]: DGCNN embedding -> mean-shift (10 iterations, q = 0.05) -> ellipsoid fit -> convex loss, forward and
    backward, B = 2 x 2048."""
    g = golden("step_dgcnn_selfsup")
    d = C.selfsup_inputs(g)
    net = C.selfsup_state(g, orc.OracleDGCNGn)
    emb, _ = net(d["xyz"])
    total, chamfer, params, labels = orc.convex_loss(d["xyz"], d["cham"], emb.permute(0, 2, 1), quantile=C.Q, iterations=C.ITERS,
                                                     max_num_clusters=25, rand_table=[[d["R"]] * 64] * 2, canonical=True,
                                                     center_ids=d["center_ids"])
    torch.mean(total).backward()
    grads = {k: (None if p.grad is None else p.grad.detach()) for k, p in net.named_parameters()}
    # (the fixture's own bars: even the same fp32 arithmetic with another thread count flips kNN neighbours, dgcnn_common)
    C.check_selfsup(g, total.detach(), chamfer.detach(), params, labels, emb.detach(), grads)
