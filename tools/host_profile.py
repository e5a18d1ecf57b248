"""Host-side profile of the step's enqueue path: cProfile over a few c3 steps (after warm-up), top functions by own time and
by cumulative time.  usage (GPU box): python tools/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
import bench   # noqa: E402


class A:
    workload, graph, ms_split = os.environ.get("HOSTPROF_WORKLOAD", "c3"), False, "0"


ctx = {"world": 1, "rank": 0, "device": torch.device("cuda", 0), "use_dist": False}
torch.cuda.set_device(0)
# borrow bench's step by running its measurement with a hook: _measure builds `step` as a closure, so re-create it here
steps = []
orig = bench.launch_census


def grab(step, path):
    steps.append(step)


bench.launch_census = grab
os.environ["PRIFIT_BENCH_CENSUS"] = "x"
bench.measure(A, ctx, "blobs", os.environ.get("HOSTPROF_EMBEDDING", "clustered"), 5, 5, full=False)
step = steps[0]
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 10
import time
t0 = time.perf_counter()
pr.enable()
for _ in range(N):
    torch.cuda.synchronize()
    step()
pr.disable()
torch.cuda.synchronize()
print("steps", N)
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    print(s.getvalue()[:9000])
