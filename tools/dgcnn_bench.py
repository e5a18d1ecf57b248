"""configs[4]: DGCNN (k=20) + mean-shift + ellipsoid fit + convex loss at B=24 x 2048, fwd+bwd+Adam (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
from prifit_amd.src import dgcnn as D
from prifit_amd.train_step import SpeculativeRunner

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = D.get_model(50, k=20).to(dev).train()
data = bench.make_inputs("c3", 0, dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
runner = SpeculativeRunner(net)

def fwd_bwd():
    out = net(data["xyz"], None, chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25)
    loss = out[3].mean(); loss.backward(); return loss

def step():
    opt.zero_grad(set_to_none=True)
    loss = runner.run(fwd_bwd, lambda: opt.zero_grad(set_to_none=True))
    opt.step(); return loss

for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("dgcnn c5: %.2f ms/step  %.1f shapes/s  loss %.5f fallbacks %d" % (dt * 1e3, 24 / dt, float(loss), runner.fallbacks))
