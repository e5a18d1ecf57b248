import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
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
ts = []
for i in range(70):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bucket.zero(); runner.run(fb, bucket.zero); bucket.allreduce(); opt.step()
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join("%.1f" % t for t in ts))
print("mem reserved GB", torch.cuda.memory_reserved() / 1e9)
