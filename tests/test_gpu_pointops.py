"""GPU parity (through the C ABI): HIP index / grouping kernels vs the CPU oracle and the goldens.

Bit-exact for every integer result (FPS, ball query, 3-NN indices) and for the expanded-form
distances; fp32 tolerance 1e-6 for interpolation / gather (pure data movement + 3 FMAs)."""
import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops(hiplib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from prifit_amd import ops as o
    return o


@pytest.mark.parametrize("kind,N", [("cube", 2048), ("surface", 2048), ("cube", 1024), ("surface", 1024)])
def test_index_ops_vs_golden_and_oracle(ops, golden, kind, N):
    g = golden(f"index_{kind}_n{N}")
    seed, B = int(g["seed"]), 4
    xyz_c = _t(synth.cloud(kind, B, N, seed))
    xyz = xyz_c.cuda()
    f1, c1 = ops.farthest_point_sample(xyz, 512, _t(g["start1"]).cuda(), return_xyz=True)
    assert torch.equal(f1.cpu(), _t(g["fps1"]).long())
    assert torch.equal(c1.cpu(), orc.gather_rows(xyz_c, f1.cpu()))
    f2, c2 = ops.farthest_point_sample(c1, 128, _t(g["start2"]).cuda(), return_xyz=True)
    assert torch.equal(f2.cpu(), _t(g["fps2"]).long())
    # all MSG radii of a layer in ONE launch, int32 and int64 outputs
    for lname, pts, ctr, rs, ks in (("sa1", xyz, c1, [0.1, 0.2, 0.4, 0.2], [32, 64, 128, 32]),
                                    ("sa2", c1, c2, [0.4, 0.8], [64, 128])):
        for idx64 in (False, True):
            outs = ops.ball_query_multi(rs, ks, pts, ctr, idx64=idx64)
            for r, k, gi in zip(rs, ks, outs):
                gi = gi.cpu().long()
                assert torch.equal(gi, orc.c_query_ball_point(r, k, pts.cpu(), ctr.cpu())), (lname, r)
                assert torch.equal(gi[:, :64], _t(g[f"ball_{lname}_{r}_{k}_head"]).long())
                assert torch.equal(gi.sum(dim=(1, 2)), _t(g[f"ball_{lname}_{r}_{k}_sum"]))
    for lname, a, b in (("fp1", xyz, c1), ("fp2", c1, c2)):
        idx, w, d = ops.three_nn(a, b, want_dist=True)
        assert torch.equal(idx.cpu().long(), _t(g[f"nn3_{lname}_idx"]).long())
        assert torch.equal(d.cpu(), _t(g[f"nn3_{lname}_d"]))
        recip = 1.0 / (d.cpu() + 1e-8)
        torch.testing.assert_close(w.cpu(), recip / recip.sum(-1, keepdim=True), rtol=1e-6, atol=1e-7)
    assert torch.equal(ops.square_distance(c1, c2).cpu()[:2, :128], _t(g["sqdist_fp2_head"]))
    assert torch.equal(ops.square_distance(xyz, c1).cpu(), orc.c_square_distance(xyz_c, c1.cpu()))


def test_fps_ragged_sizes_and_ties(ops):
    # N not a multiple of 64/256, duplicated points (ties -> lowest index), npoint == N
    for N, S, seed in ((1, 1, 0), (63, 17, 1), (257, 257, 2), (1000, 333, 3), (4096, 64, 4)):
        xyz = _t(synth.cloud("cube", 3, N, seed))
        if N > 10:
            xyz[:, N // 2:] = xyz[:, : N - N // 2]  # exact duplicates
        st = _t(synth.fps_start(3, N, seed))
        ref = orc.c_farthest_point_sample(xyz, S, st)
        got = ops.farthest_point_sample(xyz.cuda(), S, st.cuda())
        assert torch.equal(got.cpu(), ref), N


def test_ball_query_edges(ops):
    # empty balls keep N (reference quirk), overfull balls truncate in index order, S not multiple of 16
    xyz = _t(synth.cloud("cube", 2, 300, 5))
    q = torch.cat([xyz[:, :37], torch.full((2, 3, 3), 9.0)], dim=1)
    for r, k in ((0.05, 8), (0.5, 16), (3.0, 64)):
        ref = orc.c_query_ball_point(r, k, xyz, q)
        got = ops.ball_query_multi([r], [k], xyz.cuda(), q.cuda(), idx64=True)[0]
        assert torch.equal(got.cpu(), ref), r
    # N larger than one LDS tile
    xyz = _t(synth.cloud("surface", 1, 5000, 6))
    q = xyz[:, ::50].contiguous()
    ref = orc.c_query_ball_point(0.3, 128, xyz, q)
    got = ops.ball_query_multi([0.3], [128], xyz.cuda(), q.cuda(), idx64=True)[0]
    assert torch.equal(got.cpu(), ref)


def test_three_nn_large_and_ties(ops):
    xyz2 = _t(synth.cloud("cube", 2, 2500, 7))
    xyz2[:, 100:200] = xyz2[:, :100]  # duplicates: ties resolve to the lower index
    xyz1 = _t(synth.cloud("cube", 2, 777, 8))
    d, i = orc.c_three_nn(xyz1, xyz2)
    idx, w, dist = ops.three_nn(xyz1.cuda(), xyz2.cuda(), want_dist=True)
    assert torch.equal(idx.cpu().long(), i) and torch.equal(dist.cpu(), d)


@pytest.mark.parametrize("C,order", [(0, 0), (3, 0), (16, 0), (320, 0), (16, 1), (7, 1)])
def test_group_gather_and_scatter(ops, C, order):
    B, N, S, K = 2, 512, 64, 16
    xyz = _t(synth.cloud("surface", B, N, 9))
    feat = _t(synth.features(B, N, C, 9)) if C else None
    st = _t(synth.fps_start(B, N, 9))
    ctr = orc.gather_rows(xyz, orc.c_farthest_point_sample(xyz, S, st))
    gi = orc.c_query_ball_point(0.3, K, xyz, ctr)
    rel = orc.gather_rows(xyz, gi) - ctr.unsqueeze(2)
    if C:
        gf = orc.gather_rows(feat, gi)
        ref = torch.cat([gf, rel], -1) if order == 0 else torch.cat([rel, gf], -1)
    else:
        ref = rel
    gi32 = gi.int().cuda()
    out = ops.group_gather(None if feat is None else feat.cuda(), xyz.cuda(), ctr.cuda(), gi32, order=order)
    ld = out.shape[1]
    assert ld % 4 == 0 and ld >= C + 3
    assert torch.equal(out[:, :C + 3].cpu(), ref.reshape(-1, C + 3))
    assert out[:, C + 3:].abs().max().item() == 0 if ld > C + 3 else True
    if C:
        gout = _t(synth.features(1, B * S * K, ld, 10))[0].cuda()
        col0 = 0 if order == 0 else 3
        dfeat = ops.group_scatter_add(gout, col0, gi32, B, N, C)
        ref_d = torch.zeros(B, N, C)
        ref_d.scatter_add_(1, gi.reshape(B, -1, 1).expand(-1, -1, C),
                           gout[:, col0:col0 + C].cpu().reshape(B, S * K, C))
        torch.testing.assert_close(dfeat.cpu(), ref_d, rtol=1e-5, atol=1e-5)


def test_three_interpolate_fwd_bwd(ops):
    B, N, S, C = 2, 700, 90, 24
    xyz1 = _t(synth.cloud("cube", B, N, 11))
    xyz2 = _t(synth.cloud("cube", B, S, 12))
    p2 = _t(synth.features(B, S, C, 11)).requires_grad_(True)
    d3, i3 = orc.c_three_nn(xyz1, xyz2)
    ref = orc.three_interpolate(p2, d3, i3)
    idx, w = ops.three_nn(xyz1.cuda(), xyz2.cuda())
    buf = torch.zeros(B * N, C + 8, device="cuda")
    ops.three_interpolate(p2.detach().cuda(), idx, w, out=buf, col0=8)
    torch.testing.assert_close(buf[:, 8:].cpu().reshape(B, N, C), ref.detach(), rtol=1e-5, atol=1e-6)
    assert buf[:, :8].abs().max().item() == 0
    g = _t(synth.features(B, N, C, 13))
    (ref * g).sum().backward()
    gb = torch.zeros(B * N, C + 8, device="cuda")
    gb[:, 8:] = g.reshape(B * N, C).cuda()
    dp2 = ops.three_interpolate_bwd(gb, 8, idx, w, B, S, C)
    torch.testing.assert_close(dp2.cpu(), p2.grad, rtol=1e-4, atol=1e-5)


def test_cur_stream_is_the_current_stream(hiplib):
    """_lib.cur_stream (the raw accessor) names the same hipStream_t as torch.cuda.current_stream(), on the default stream,
    inside a stream context, and for an explicit device."""
    from prifit_amd._lib import cur_stream
    assert (cur_stream() or 0) == torch.cuda.current_stream().cuda_stream        # (a plain int: argtypes converts it to void *)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert cur_stream() == side.cuda_stream and cur_stream(torch.device("cuda", 0)) == side.cuda_stream
        assert cur_stream("cuda") == side.cuda_stream and cur_stream(0) == side.cuda_stream
    assert (cur_stream() or 0) == torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("B,N,S,D2,D1", [(2, 700, 90, 24, 10), (3, 256, 1, 16, 6), (2, 128, 40, 7, 0), (1, 64, 33, 256, 128)])
def test_fp_rows_is_interpolate_concat_pad(ops, B, N, S, D2, D1):
    """prifit_fp_rows (round 6): [interpolated | points1 | 0-pad] rows of a feature-propagation layer in one launch against the
    three steps it replaces (prifit_three_interpolate / the S == 1 broadcast of models/pointnet_util.py:287-288, torch.cat,
    zero padding), forward bit for bit (the same products in the same order) and backward through autograd."""
    from prifit_amd.nn_ops import FpRowsFn, ThreeInterpolateFn
    gen = torch.Generator().manual_seed(B * N + S)
    xyz1 = _t(synth.cloud("cube", B, N, 3)).cuda()
    xyz2 = xyz1[:, :S].contiguous() if S > 1 else xyz1[:, :1].contiguous()
    p2 = torch.randn(B, S, D2, generator=gen).cuda().requires_grad_(True)
    p1 = torch.randn(B, N, D1, generator=gen).cuda().requires_grad_(True) if D1 else None
    kp = (D1 + D2 + 3) // 4 * 4
    idx, w = ops.three_nn(xyz1, xyz2) if S > 1 else (None, None)
    rows = FpRowsFn.apply(p2, idx, w, p1, kp)
    p2r = p2.detach().clone().requires_grad_(True)
    p1r = p1.detach().clone().requires_grad_(True) if D1 else None
    interp = ThreeInterpolateFn.apply(p2r, idx, w) if S > 1 else p2r.expand(B, N, D2).reshape(B * N, D2)
    parts = [interp] + ([p1r.reshape(B * N, D1)] if D1 else []) + ([torch.zeros(B * N, kp - D1 - D2, device="cuda")] if kp > D1 + D2 else [])
    ref = torch.cat(parts, dim=1)
    assert rows.shape == (B * N, kp) and torch.equal(rows, ref)
    G = torch.randn(B * N, kp, generator=gen).cuda()
    (rows * G).sum().backward()
    (ref * G).sum().backward()
    torch.testing.assert_close(p2.grad, p2r.grad, rtol=1e-5, atol=1e-5)      # (atomics in the scatter: rounding order)
    if D1:
        assert torch.equal(p1.grad, p1r.grad)


def test_pack_cols_multi_equals_single_launches(hiplib):
    """prifit_pack_cols_multi / prifit_unpack_cols_multi (round 6): several column-packed weights in one launch per direction
    against PackColsFn one by one -- copies and fixed-order sums: bit for bit; maps with zero columns and repeated sources."""
    from prifit_amd.models.pointnet_util import PackAllFn, PackColsFn
    gen = torch.Generator().manual_seed(4)
    shapes = [(64, 6), (128, 131), (196, 323), (16, 3), (256, 515)]
    maps = [(3, 4, 5, 0, 1, 2, -1, -1), tuple(range(3, 131)) + (0, 1, 2) + (-1,), tuple(range(3, 323)) + (0, 1, 2, -1),
            (0, 0, 1, 2), tuple(range(512, 515)) + tuple(range(0, 512)) + (-1,)]
    ws = [torch.randn(s, generator=gen).cuda().requires_grad_(True) for s in shapes]
    wr = [w.detach().clone().requires_grad_(True) for w in ws]
    outs = PackAllFn.apply(tuple(maps), *ws)
    refs = [PackColsFn.apply(w, m) for w, m in zip(wr, maps)]
    Gs = [torch.randn(o.shape, generator=gen).cuda() for o in outs]
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    sum((o * g).sum() for o, g in zip(outs[:-1], Gs[:-1])).backward()       # (the last output unused: its job is skipped)
    sum((r * g).sum() for r, g in zip(refs[:-1], Gs[:-1])).backward()
    for a, b in zip(ws[:-1], wr[:-1]):
        assert torch.equal(a.grad, b.grad)
    assert ws[-1].grad is None
