"""Which python call sites launch the small torch kernels (fills, copies, elementwise) of a C3 step?  (GPU box)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
net, M = bench.build_model(dev)
data = bench.make_inputs("c3", 0, dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
from prifit_amd.train_step import SpeculativeRunner
runner = SpeculativeRunner(net)
def fb():
    out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25, fps_start=(data["s1"], data["s2"]))
    loss = out[3].mean(); loss.backward(); return loss
def full():
    for p in net.parameters(): p.grad = None
    runner.run(fb, lambda: None); opt.step()
for _ in range(3): full()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    full(); torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_stack_n=12):
    t = getattr(ka, "self_device_time_total", 0) or getattr(ka, "self_cuda_time_total", 0)
    if t <= 0 or not ka.key.startswith("aten::"):
        continue
    site = "?"
    for fr in ka.stack:
        if "/prifit_amd/" in fr or "bench.py" in fr:
            site = fr.split("/prifit_amd/")[-1] if "/prifit_amd/" in fr else fr.split("/")[-1]
            break
    rows.append((t, ka.count, ka.key, site))
rows.sort(reverse=True)
for t, n, name, site in rows[:70]:
    print("%8.1f us  x%-3d %-30s %s" % (t, n, name, site[:110]))
print("total aten self device time %.1f us in %d calls" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
