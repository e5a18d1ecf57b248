"""Which python call sites produce the zero-fill kernels of a C3 step?  (GPU box; counts torch.zeros / zeros_like /
new_zeros / zero_ / masked_fill calls per site for one step)"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
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
sites = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/prifit_amd/" in fr.filename or fr.filename.endswith("fill_sites.py") or fr.filename.endswith("bench.py"):
            return "%s:%d %s" % (fr.filename.split("/prifit_amd/")[-1].split("/")[-1], fr.lineno, fr.name)
    return "torch-internal (autograd engine / optimizer)"
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for obj, name in ((torch, "zeros"), (torch, "zeros_like"), (torch, "full"), (torch.Tensor, "new_zeros"), (torch.Tensor, "zero_"),
                  (torch.Tensor, "fill_"), (torch.Tensor, "masked_fill"), (torch, "ones"), (torch, "ones_like"),
                  # copies / concatenations (the other third of the small torch launches)
                  (torch, "cat"), (torch, "stack"), (torch.Tensor, "contiguous"), (torch.Tensor, "clone"), (torch.Tensor, "copy_"),
                  (torch.Tensor, "float"), (torch.Tensor, "sum"), (torch, "sum")):
    wrap(obj, name)
full(); torch.cuda.synchronize()
for (name, s), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%3d  %-12s %s" % (n, name, s))
print("total", sum(sites.values()))
