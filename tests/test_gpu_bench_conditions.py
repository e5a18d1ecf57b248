"""GPU: the training-like benchmark condition (`bench.py --embedding clustered`, reported under `extra` beside the headline)
is the reference's arithmetic too -- on the full B = 24 x 2048 workload the clusters per shape, the label partition and
the per-shape loss of 2 of the 24 shapes agree with the oracle (README.md:59-64 regime: several clusters per shape,
quantile 0.05, 10 mean-shift iterations, <= 25 clusters; convex_loss.py:27-103)."""
import os
import sys

import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _same_partition(la, lb):
    pairs = torch.unique(torch.stack([la.long(), lb.long()], 1), dim=0)
    return pairs.shape[0] == torch.unique(la).shape[0] == torch.unique(lb).shape[0]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("cloud", ["blobs", "surface"])
def test_clustered_embedding_condition_matches_oracle_on_two_shapes(hiplib, cloud):
    sys.path.insert(0, ROOT)
    import bench
    from prifit_amd.convex_loss import convex_loss
    dev = torch.device("cuda", 0)
    net, _ = bench.build_model(dev, "c3")
    data = bench.make_inputs("c3", 0, dev, cloud)
    starts = (data["s1"], data["s2"])
    off = torch.from_numpy(synth.part_embedding_offset(data["parts"].cpu().numpy(), 128, 0)).to(dev)
    net.train()
    R = torch.from_numpy(synth.uniform01((3, 3), 11))
    kw = dict(quantile=0.05, max_num_clusters=25)
    with torch.no_grad():
        out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, msc_iterations=10,
                  fps_start=starts, fit_inputs=dict(rand_table=R.to(dev), canonical=True, embedding_offset=off), **kw)
        emb = out[7].detach().contiguous()                           # [B,128,N]: the head's output, before the offset
        total, l, params, labels, info = convex_loss(data["xyz"], data["chamfer"], emb, iterations=10, rand_table=R.to(dev),
                                                     canonical=True, return_info=True, embedding_offset=off, **kw)
    K = info["cluster"]["count"].cpu().tolist()
    # ~8 clusters per shape (a Voronoi cell of a few points on a surface cloud merges with its neighbour)
    assert len(K) == 24 and min(K) >= 3 and max(K) <= 10 and sum(K) / 24.0 >= 7, K
    assert torch.allclose(total.cpu(), out[3].cpu(), rtol=1e-5, atol=1e-7)                 # the network ran this very loss
    per_shape = ((info["parts"][0] + info["parts"][1]) / 2.0).cpu()
    for b in (3, 17):
        Xo = (emb[b:b + 1] + off[b:b + 1].permute(0, 2, 1)).cpu()
        t_o, _, params_o, labels_o, info_o = orc.convex_loss(
            data["xyz"][b:b + 1].cpu(), data["chamfer"][b:b + 1].cpu(), Xo, iterations=10,
            rand_table=[[R] * 64], canonical=True, return_info=True, **kw)
        assert len(params_o[0]) == int(info["valid"][b].sum()) and info_o["W"][0].shape[1] == K[b]
        assert _same_partition(labels[b].cpu(), labels_o[0])
        # the bar: 1e-4 (north star), or -- where the fit itself is ill-conditioned in fp32, e.g. the flat patches of a
        # surface cloud -- 3 x the distance between the oracle in fp32 and in fp64 on the same inputs (measured, not chosen)
        p64 = orc.convex_loss(data["xyz"][b:b + 1].cpu().double(), data["chamfer"][b:b + 1].cpu().double(), Xo.double(),
                              iterations=10, rand_table=[[R.double()] * 64], canonical=True, return_info=True, **kw)[4]["parts"][0]
        t_64 = (p64[0] + p64[1]) / 2.0
        bar = max(1e-4, 3.0 * abs(float(t_o) - float(t_64)) / abs(float(t_64)))
        assert abs(float(per_shape[b]) - float(t_o)) <= bar * abs(float(t_o)) + 1e-7, (b, float(per_shape[b]), float(t_o), float(t_64), bar)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("embedding", ["clustered25", "retry40"])
def test_loaded_fit_path_conditions_match_oracle_on_two_shapes(hiplib, embedding):
    """The two conditions that LOAD the fit path (`bench.py --embedding clustered25`: 25 equal-size parts, K at the
    `max_num_clusters` cap; `--embedding retry40`: 40 tight parts at quantile 0.01, so that guard_mean_shift's quantile-doubling
    retry runs, src/ellipsoid_utils.py:19-27) on the full B = 24 workload: clusters per shape, the label partition and the
    per-shape loss of two shapes against the oracle -- through the synchronous path, retries included."""
    sys.path.insert(0, ROOT)
    import bench
    from prifit_amd.convex_loss import convex_loss
    dev = torch.device("cuda", 0)
    net, _ = bench.build_model(dev, "c3")
    data = bench.make_inputs("c3", 0, dev, "blobs")
    (nx, ny), noise, q = bench.EMBEDDING_PARTS[embedding]
    parts = synth.equal_part_labels(data["xyz"].transpose(1, 2).cpu().numpy(), nx, ny)
    off = torch.from_numpy(synth.part_embedding_offset(parts, 128, 0, K=nx * ny, noise=noise, scale=100.0)).to(dev)
    net.train()
    R = torch.from_numpy(synth.uniform01((3, 3), 11))
    kw = dict(quantile=q, max_num_clusters=25)
    with torch.no_grad():
        out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, msc_iterations=10,
                  fps_start=(data["s1"], data["s2"]), fit_inputs=dict(rand_table=R.to(dev), canonical=True, embedding_offset=off), **kw)
        emb = out[7].detach().contiguous()
        total, l, params, labels, info = convex_loss(data["xyz"], data["chamfer"], emb, iterations=10, rand_table=R.to(dev),
                                                     canonical=True, return_info=True, embedding_offset=off, **kw)
    K = info["cluster"]["count"].cpu().tolist()
    if embedding == "clustered25":
        assert K == [25] * 24, K                                     # every shape at the cap, no retry
    else:
        assert max(K) <= 25 and min(K) >= 1, K                       # 40 modes at q = 0.01 and 0.02, then the bandwidth swallows them
    per_shape = ((info["parts"][0] + info["parts"][1]) / 2.0).cpu()
    for b in (3, 17):
        Xo = (emb[b:b + 1] + off[b:b + 1].permute(0, 2, 1)).cpu()
        t_o, _, params_o, labels_o, info_o = orc.convex_loss(
            data["xyz"][b:b + 1].cpu(), data["chamfer"][b:b + 1].cpu(), Xo, iterations=10,
            rand_table=[[R] * 64], canonical=True, return_info=True, **kw)
        assert info_o["W"][0].shape[1] == K[b], (info_o["W"][0].shape, K[b])
        if embedding == "retry40":
            assert info_o["cluster"][0]["quantile"] > q               # the oracle took the retry too
        assert _same_partition(labels[b].cpu(), labels_o[0])
        p64 = orc.convex_loss(data["xyz"][b:b + 1].cpu().double(), data["chamfer"][b:b + 1].cpu().double(), Xo.double(),
                              iterations=10, rand_table=[[R.double()] * 64], canonical=True, return_info=True, **kw)[4]["parts"][0]
        t_64 = (p64[0] + p64[1]) / 2.0
        bar = max(1e-4, 3.0 * abs(float(t_o) - float(t_64)) / abs(float(t_64)))
        assert abs(float(per_shape[b]) - float(t_o)) <= bar * abs(float(t_o)) + 1e-7, (b, float(per_shape[b]), float(t_o), float(t_64), bar)
