"""Which python call sites issue the small torch kernels of a c3 step?  (GPU box)  Records every torch function called
during one step (TorchFunctionMode) with the nearest prifit_amd / bench call site, and prints the counts."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.overrides import TorchFunctionMode
import bench
dev = torch.device("cuda", 0)
net, M = bench.build_model(dev)
data = bench.make_inputs("c3", 0, dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
from prifit_amd.train_step import SpeculativeRunner
from prifit_amd.ddp import FlatGradBucket
runner = SpeculativeRunner(net)
bucket = FlatGradBucket(net)
def fb():
    out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25, fps_start=(data["s1"], data["s2"]))
    loss = out[3].mean(); loss.backward(); return loss
def full():
    bucket.zero(); runner.run(fb, bucket.zero); bucket.allreduce(); opt.step()
for _ in range(2): full()
sites = collections.Counter()
SKIP = {"size", "dim", "is_floating_point", "shape", "__get__", "stride", "data_ptr", "is_contiguous", "numel", "view", "reshape",
        "permute", "transpose", "unsqueeze", "expand", "unbind", "detach", "requires_grad_", "view_as", "t", "__getitem__", "squeeze",
        "is_cuda", "device", "dtype", "contiguous", "apply", "backward", "grad", "__set__", "__setitem__"}
class Rec(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name not in SKIP:
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/prifit_amd/" in fr.filename or fr.filename.endswith("glue_sites.py"):
                    site = "%s:%d" % (fr.filename.split("/")[-1], fr.lineno); break
            sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Rec():
    full()
torch.cuda.synchronize()
for (name, s), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print("%3d  %-22s %s" % (n, name, s))
