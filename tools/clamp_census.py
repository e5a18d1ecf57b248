"""How much of the mean-shift kernel matrix K = exp(clamp((Z X^T - 1)/b^2, -13, 75)) sits on the lower clamp?  (GPU box)
Where the clamp is active the gradient is exactly zero (src/guard.py:6-11), so a 32 x 32 tile of K that is clamped
everywhere would let the backward skip that tile in all four of its products.  Reports, for the benchmark's network
embedding and for the prototype embeddings of the fit fixtures, per iteration: fraction of clamped ENTRIES and fraction of
fully clamped 32 x 32 TILES in the natural point order and with the points sorted by their final cluster label."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
import bench
from prifit_amd import fit_ops, synth

dev = torch.device("cuda", 0)


def census(X, name, q=0.05, iters=10):
    with torch.no_grad():
        bw = fit_ops.compute_bandwidth(X, q)
        Z = X.clone()
        rows = []
        cl = fit_ops.cluster(X, q, iters, 25)
        order = torch.argsort(cl["labels"], dim=1, stable=True)
        for it in range(iters):
            S = torch.bmm(Z, X.transpose(1, 2))
            e = (S - 1.0) / (bw * bw).view(-1, 1, 1)
            clamped = e <= -13.0
            B, N, _ = clamped.shape

            def tiles(c):
                t = c.view(B, N // 32, 32, N // 32, 32).all(dim=4).all(dim=2)
                return t.float().mean().item()
            srt = torch.gather(torch.gather(clamped, 1, order.unsqueeze(-1).expand(-1, -1, N)), 2, order.unsqueeze(1).expand(-1, N, -1))
            rows.append((it, clamped.float().mean().item(), tiles(clamped), tiles(srt)))
            Z = fit_ops.MeanShiftFn.apply(Z, bw, 1) if False else F.normalize(torch.bmm(torch.exp(e.clamp(-13, 75)), X) /
                                                                            torch.exp(e.clamp(-13, 75)).sum(2, keepdim=True), dim=2)
        print("%s: bandwidth %.3f..%.3f, clusters %s" % (name, bw.min().item(), bw.max().item(), cl["count"].tolist()[:8]))
        for it, a, b, c in rows:
            if it in (0, 1, 4, 9):
                print("   iteration %2d: clamped entries %.3f   fully clamped 32x32 tiles: natural order %.3f, sorted by label %.3f" % (it + 1, a, b, c))


# (a) the benchmark: seeded untrained network on the blob clouds
net, M = bench.build_model(dev, "c3")
data = bench.make_inputs("c3", 0, dev)
with torch.no_grad():
    out = net(data["xyz"][:8], data["cls"][:8], embed=True, fps_start=(data["s1"][:8], data["s2"][:8]))
    emb = F.normalize(out[7].permute(0, 2, 1), dim=2).contiguous()
census(emb, "benchmark network embedding (B=8)")
# (b) prototype embeddings (fit fixtures / SURVEY 8d)
for noise in (0.01, 0.03):
    cham, lab = synth.blobs_with_labels(8, 2048, 3)
    X = torch.from_numpy(synth.prototype_embedding(lab, 128, 5, noise=noise)).to(dev)
    census(X, "prototype embedding, noise %.2f (B=8)" % noise)
