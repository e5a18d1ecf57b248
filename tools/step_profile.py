"""Where does a C3 step spend its wall time?  Sections timed with synchronize() around them (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
from prifit_amd import fit_ops
from prifit_amd.convex_loss import analytic_chamfer_distance
import torch.nn.functional as F

dev = torch.device("cuda", 0)
net, M = bench.build_model(dev)
data = bench.make_inputs("c3", 0, dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)

def T():
    torch.cuda.synchronize(); return time.perf_counter()

for it in range(4):
    t = [T()]
    opt.zero_grad(set_to_none=True)
    l1, l2, l3, feat = net.embed_features(data["xyz"], data["cls"], (data["s1"], data["s2"])); t.append(T())
    from prifit_amd.nn_ops import LinearFn
    emb = LinearFn.apply(feat, net.extra_conv_emb.weight.reshape(128, 128), net.extra_conv_emb.bias)
    X = F.normalize(emb.reshape(24, 2048, 128), dim=2).contiguous(); t.append(T())
    with torch.no_grad():
        bw = fit_ops.compute_bandwidth(X, 0.05); t.append(T())
    Z = fit_ops.MeanShiftFn.apply(X, bw, 10); t.append(T())
    with torch.no_grad():
        ids, count, labels, used = fit_ops.nms(Z.detach(), bw); t.append(T())
    cl_t0 = T()
    centres = torch.gather(Z, 1, ids[:, :fit_ops.KM].long().unsqueeze(-1).expand(-1, -1, 128))
    W = fit_ops.MembershipFn.apply(centres, X, bw, count); t.append(T())
    pts = data["xyz"].permute(0, 2, 1).contiguous()
    r, V, c, valid = fit_ops.EllipsoidFitFn.apply(pts, W, count, torch.rand(24, 32, 3, 3, device=dev), True); t.append(T())
    loss, _ = analytic_chamfer_distance(r, V, c, valid, data["chamfer"].permute(0, 2, 1).contiguous()); t.append(T())
    loss.backward(); t.append(T())
    opt.step(); t.append(T())
    names = ["backbone fwd", "emb+norm", "bandwidth", "meanshift fwd", "nms", "membership", "fit", "chamfer", "backward", "adam"]
    if it >= 2:
        print(" | ".join("%s %.2f" % (n, 1e3 * (b - a)) for n, a, b in zip(names, t[:-1], t[1:])), "| total %.2f ms" % (1e3 * (t[-1] - t[0])), "K", count.tolist()[:6])

# host-only enqueue time of a full step (no syncs except the one inside cluster())
def full():
    opt.zero_grad(set_to_none=True)
    out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25, fps_start=(data["s1"], data["s2"]))
    out[3].mean().backward(); opt.step()
for _ in range(2): full()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): full()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("full step: host-side %.2f ms/step, +drain %.2f ms" % (1e3 * (t1 - t0) / 5, 1e3 * (t2 - t1)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); full(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
