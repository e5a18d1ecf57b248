"""Host time of one step by phase (forward / backward / verdict wait / exchange + optimizer), from an empty queue.
usage (GPU box): python tools/host_split.py [c3|c5|c2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from prifit_amd import fit_ops, train_step

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
T = {"fwd": [], "bwd": [], "wait": [], "opt": []}
marks = {}

class A:
    workload, graph, ms_split, default_condition = w, False, "0", True

orig_run = train_step.SpeculativeRunner.run
def run(self, fn, reset):
    # same as SpeculativeRunner.run, with the verdict wait timed apart
    for dst, src in zip(self.snap, self.bufs):
        torch._foreach_copy_(dst, src)
    with fit_ops.speculative() as spec:
        out = fn()
    marks["enq"] = time.perf_counter()
    ok = spec.ok()
    marks["ok"] = time.perf_counter()
    assert ok
    return out
train_step.SpeculativeRunner.run = run
ob = torch.Tensor.backward
def backward(self, *a, **k):
    marks["b0"] = time.perf_counter()
    r = ob(self, *a, **k)
    marks["b1"] = time.perf_counter()
    return r
torch.Tensor.backward = backward

steps = []
bench.launch_census = lambda step, path: steps.append(step)
os.environ["PRIFIT_BENCH_CENSUS"] = "x"
ctx = {"world": 1, "rank": 0, "device": torch.device("cuda", 0), "use_dist": False}
torch.cuda.set_device(0)
bench.measure(A, ctx, bench.DEFAULT_CLOUD[w], bench.DEFAULT_EMBEDDING[w], 5, 5, full=False)
step = steps[0]
import gc; gc.collect(); gc.disable()
for i in range(25):
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    if i < 5: continue
    b0, b1 = marks["b0"], marks["b1"]
    enq, ok = marks.get("enq", b1), marks.get("ok", b1)
    T["fwd"].append(b0 - t0); T["bwd"].append(b1 - b0); T["wait"].append(ok - enq); T["opt"].append(t1 - ok + (enq - b1))
torch.cuda.synchronize()
import statistics as st
tot = 0
for k, v in T.items():
    m = st.median(v) * 1e3; tot += m if k != "wait" else 0
    print("%s %-5s median %.3f ms  min %.3f  max %.3f" % (w, k, m, min(v) * 1e3, max(v) * 1e3))
print("%s host enqueue without the verdict wait: %.3f ms per step" % (w, tot))
if "--profile" in sys.argv:
    # where the Python time of a step goes: cProfile over 10 steps from an empty queue (the profiler's own overhead inflates
    # everything ~1.5-2x; the ORDER is what this is for)
    import cProfile, pstats, io
    pr = cProfile.Profile()
    for i in range(10):
        torch.cuda.synchronize()
        pr.enable()
        step()
        pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(45)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats("cumulative").print_stats(60)
    print(s.getvalue())
