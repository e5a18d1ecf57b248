import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import prifit_oracle as orc
from prifit_amd import synth, nn_ops
def _t(a): return torch.from_numpy(np.asarray(a))
import json
kind,B,N,S,D,feat_first = ("cube", 3, 1000, 37, 3, True)
radii,ks,widths = json.loads(os.environ.get("CFG", "[[0.1, 0.2, 0.4, 0.3], [32, 64, 128, 16], [32, 64, 64, 16]]"))
xyz_c = _t(synth.cloud(kind, B, N, 5)); xyz_c[:, 7] = xyz_c[:, 3]
start = torch.zeros(B, dtype=torch.long)
fps = orc.c_farthest_point_sample(xyz_c, S, start)
ctr_c = orc.gather_rows(xyz_c, fps); ctr_c[0, 0] = torch.tensor([9.0, 9.0, 9.0])
g = torch.Generator().manual_seed(1)
feat_c = xyz_c
Ws = [torch.randn(c, D + 3, generator=g) * 0.5 for c in widths]
bs = [torch.randn(c, generator=g) if i % 2 == 0 else None for i, c in enumerate(widths)]
dev = "cuda"
feat = feat_c.to(dev).contiguous(); xyz_d, ctr_d = xyz_c.to(dev), ctr_c.to(dev)
for fx in (False, True):
    Ys, slabs, idxs = nn_ops._sa_group_launch(0, xyz_d, ctr_d, feat, feat_first, radii, ks, widths, [w.to(dev) for w in Ws], None, None, [None if b is None else b.to(dev) for b in bs], feat_xyz=fx)
    torch.cuda.synchronize()
    for r, k, c, W, b, Y, idx in zip(radii, ks, widths, Ws, bs, Ys, idxs):
        want = orc.c_query_ball_point(r, k, xyz_c, ctr_c)
        okidx = torch.equal(idx.cpu().long(), want)
        rows = torch.cat([feat_c.double()[torch.arange(B)[:, None, None], want.clamp(max=N - 1)] * (want < N)[..., None], (xyz_c.double()[torch.arange(B)[:, None, None], want.clamp(max=N-1)] - ctr_c.double()[:, :, None]) * (want < N)[..., None]], -1)
        ref = rows @ W.double().t() + (b.double() if b is not None else 0)
        d = (Y.cpu().double().view(B, S, k, c) - ref).abs()
        bad = (d > 1e-4).nonzero()
        print("fx", fx, "r", r, "K", k, "C", c, "idx ok", okidx, "max err", float(d.max()), "bad", len(bad), bad[:5].tolist())
