"""Micro-benchmark of ball query + grouping at the B=24 MSG shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from prifit_amd import synth
from prifit_amd import ops
def timed(name, fn, bytes_, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    print("%-40s %8.1f us  %7.0f GB/s (alg)" % (name, us, bytes_ / us / 1e3))
B = 24
for kind in ("cube", "surface"):
    xyz = torch.from_numpy(synth.cloud(kind, B, 2048, 0)).cuda()
    _, c1 = ops.farthest_point_sample(xyz, 512, torch.zeros(B, dtype=torch.long, device="cuda"), return_xyz=True)
    _, c2 = ops.farthest_point_sample(c1, 128, torch.zeros(B, dtype=torch.long, device="cuda"), return_xyz=True)
    f1 = torch.randn(B, 512, 320, device="cuda")
    timed(kind + " ball sa1 (3 radii)", lambda: ops.ball_query_multi([.1, .2, .4], [32, 64, 128], xyz, c1), B * (12 * 2560 + 4 * 512 * 224))
    timed(kind + " ball sa2 (2 radii)", lambda: ops.ball_query_multi([.4, .8], [64, 128], c1, c2), B * (12 * 640 + 4 * 128 * 192))
    i1 = ops.ball_query_multi([.1, .2, .4], [32, 64, 128], xyz, c1)
    i2 = ops.ball_query_multi([.4, .8], [64, 128], c1, c2)
    for K, ii in zip((32, 64, 128), i1):
        timed(kind + " gather sa1 K=%d (C=3)" % K, lambda ii=ii: ops.group_gather(xyz, xyz, c1, ii, ld_out=8), B * (4 * 512 * K * 6))
    for K, ii in zip((64, 128), i2):
        timed(kind + " gather sa2 K=%d (C=320)" % K, lambda ii=ii: ops.group_gather(f1, c1, c2, ii, ld_out=324), B * (4 * 512 * 320 + 4 * 128 * K * 323))
    # first SA2 layer by linearity: gather of C1-wide projected rows (GatherLinearFn)
    from prifit_amd.nn_ops import GatherLinearFn
    U = torch.randn(B, 512, 128, device="cuda", requires_grad=True)
    for K, ii in zip((64, 128), i2):
        Vc = torch.randn(B, 128, 128, device="cuda", requires_grad=True)
        timed(kind + " gather_linear sa2 K=%d (C1=128)" % K, lambda ii=ii: GatherLinearFn.apply(U.detach(), Vc.detach(), None, ii, True),
              4 * B * (512 * 128 + 128 * 128 + 128 * K + 128 * K * 128))
        y, _ = GatherLinearFn.apply(U, Vc, None, ii, True)
        g = torch.randn_like(y)
        timed(kind + " gather_linear bwd K=%d" % K, lambda: torch.autograd.grad(y, (U, Vc), g, retain_graph=True),
              4 * B * (512 * 128 + 128 * 128 + 128 * K + 128 * K * 128))
    # one launch per set-abstraction level: ball query + grouping + first MLP layer (csrc/sa_group.hip)
    from prifit_amd import nn_ops
    Ws = [torch.randn(c, 6, device="cuda") for c in (32, 64, 64)]
    w1 = B * (12 * 2560 + sum(4 * 512 * k * (1 + c) for k, c in zip((32, 64, 128), (32, 64, 64))) + 4 * 2048 * 3)
    timed(kind + " sa_group_linear sa1 direct (3 radii)", lambda: nn_ops._sa_group_launch(0, xyz, c1, xyz, True, [.1, .2, .4], [32, 64, 128], [32, 64, 64], Ws, None, None, [None] * 3), w1)
    timed(kind + " sa_group_linear sa1 direct, feat = xyz from LDS", lambda: nn_ops._sa_group_launch(0, xyz, c1, xyz, True, [.1, .2, .4], [32, 64, 128], [32, 64, 64], Ws, None, None, [None] * 3, feat_xyz=True), w1)
    Us = [torch.randn(B, 512, 128, device="cuda") for _ in range(2)]
    Vs = [torch.randn(B, 128, 128, device="cuda") for _ in range(2)]
    w2 = B * (12 * 640 + sum(4 * 128 * k * 129 for k in (64, 128)) + 2 * 4 * 640 * 128)
    timed(kind + " sa_group_linear sa2 gather (2 radii)", lambda: nn_ops._sa_group_launch(1, c1, c2, None, True, [.4, .8], [64, 128], [128, 128], None, Us, Vs, [None] * 2), w2)
    Y, _, ii = nn_ops._sa_group_launch(0, xyz, c1, xyz, True, [.4], [128], [64], Ws[2:], None, None, [None])
    gy = torch.randn_like(Y[0])
    part = torch.empty(1024, 64, 6, device="cuda")
    timed(kind + " sa_first_layer_dw K=128 C=64", lambda: nn_ops.call("prifit_sa_first_layer_dw", nn_ops.ptr(gy), nn_ops.ptr(ii[0]), nn_ops.ptr(xyz), nn_ops.ptr(c1), nn_ops.ptr(xyz), B, 2048, 512, 128, 64, 3, 1, 1024, nn_ops.ptr(part), nn_ops.cur_stream()), 4 * B * 512 * 128 * 65)
