import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
dev = torch.device("cuda", 0)
net, M = bench.build_model(dev)
data = bench.make_inputs("c3", 0, dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
def full(sync_mid=False):
    opt.zero_grad(set_to_none=True)
    out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25, fps_start=(data["s1"], data["s2"]))
    if sync_mid: torch.cuda.synchronize()
    out[3].mean().backward(); opt.step()
for _ in range(3): full()
def timeit(name, fn, n=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-40s %.2f ms/step" % (name, 1e3 * (time.perf_counter() - t0) / n), flush=True)
timeit("free running", full)
timeit("sync per step", lambda: (full(), torch.cuda.synchronize()))
timeit("sync before backward", lambda: full(True))
timeit("free running again", full)
st = torch.cuda.memory_stats()
print("reserved GB %.2f allocated peak GB %.2f retries %d" % (st["reserved_bytes.all.peak"] / 2**30, st["allocated_bytes.all.peak"] / 2**30, st["num_alloc_retries"]))
print("segments", st["segment.all.current"], "inactive split GB %.2f" % (st["inactive_split_bytes.all.current"] / 2**30))
# backward only timing with events
import torch.autograd.profiler as P
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    full(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
