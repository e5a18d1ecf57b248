"""In-degree distribution of the DGCNN neighbour graphs on the benchmark input (GPU box): the gather passes of
csrc/edge_conv.hip walk a point's in-edges in one wave, so a hub sets their critical path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prifit_amd.src import dgcnn as D

dev = torch.device("cuda:0")
net, _ = bench.build_model(dev, "c5")
from prifit_amd import synth
B, N = 24, 2048
pts = torch.from_numpy(synth.cloud("blobs", B, N, 0)).to(dev)
enc = net.net.encoder
with torch.no_grad():
    idx1 = D._knn_cl(pts, 20)
    x1 = enc._edge_conv(pts, idx1, enc.conv1, N, D.edge_csr(idx1))
    idx2 = D._knn_cl(x1.view(B, N, -1), 20)
for name, idx in (("graph 1 (xyz)", idx1), ("graph 2 (features)", idx2)):
    offs = D.edge_csr(idx)[0].long()
    deg = offs[:, 1:] - offs[:, :-1]
    q = torch.quantile(deg.float().flatten(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev))
    print(name, "in-degree: mean %.1f  median %d  p90 %d  p99 %d  p99.9 %d  max %d  zero %.1f%%" % (
        deg.float().mean(), q[0], q[1], q[2], q[3], deg.max(), 100.0 * (deg == 0).float().mean()))
