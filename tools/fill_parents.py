"""Which operators launch the fill / copy kernels of a C3 step?  (GPU box; torch.profiler, parent chain of every
aten::fill_ / aten::zero_ / aten::copy_ CPU event, with input shapes)"""
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
for _ in range(2): full()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    full(); torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::fill_", "aten::zero_", "aten::copy_"]
cnt = collections.Counter()
for e in prof.events():
    if e.name in want and not (e.cpu_parent is not None and e.cpu_parent.name in want):
        chain = []
        p = e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        shp = str(e.input_shapes[0]) if e.input_shapes else "?"
        cnt[(e.name, shp, " < ".join(chain))] += 1
for (n, shp, ch), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%3d %-12s %-22s %s" % (c, n, shp, ch))
print("total", sum(cnt.values()))
