"""Per-step wall time of the c3 training step over a long run (GPU box): which steps are outliers, and whether they are the
speculation fall-backs (SpeculativeRunner re-runs a step whose cluster-count verdict asked for the quantile-doubling retry)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, bench
from prifit_amd.ddp import FlatGradBucket
from prifit_amd.train_step import SpeculativeRunner
dev = torch.device("cuda", 0)
net, M = bench.build_model(dev)
bucket = FlatGradBucket(net)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, fused=True)
data = bench.make_inputs("c3", 0, dev)
runner = SpeculativeRunner(net)
def fb():
    out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05, msc_iterations=10, max_num_clusters=25, fps_start=(data["s1"], data["s2"]))
    loss = out[3].mean(); loss.backward(); return loss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 230
import gc
gc.collect(); gc.disable()      # as bench.py does around its timed region
ts, fbs = [], []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f0 = runner.fallbacks
    bucket.zero(); runner.run(fb, bucket.zero); bucket.allreduce(); opt.step()
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0)); fbs.append(runner.fallbacks - f0)
med = sorted(ts)[len(ts) // 2]
print("median %.2f ms; fall-backs at steps %s" % (med, [i for i, f in enumerate(fbs) if f]))
print("outliers (> 1.3 x median):", [(i, round(t, 1), fbs[i]) for i, t in enumerate(ts) if t > 1.3 * med])
print("mem reserved GB", torch.cuda.memory_reserved() / 1e9)
