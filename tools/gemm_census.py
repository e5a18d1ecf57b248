"""Every GEMM launch of one C3 step with its shape, HIP-event duration, TFLOP/s and streamed GB/s (GPU box)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import bench
from prifit_amd import nn_ops, fit_ops

dev = torch.device("cuda", 0)
net, M = bench.build_model(dev)
data = bench.make_inputs("c3", 0, dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
records = []
orig = nn_ops.gemm
def traced(layout, M_, N_, K_, A, lda, B, ldb, C, ldc, batch=1, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(layout, M_, N_, K_, A, lda, B, ldb, C, ldc, batch=batch, **kw); e.record()
    flags = "".join(k[0] for k in ("a_affine", "b_affine", "bias", "stats", "aux", "a_rowsum") if kw.get(k) is not None)
    records.append((("NT", "NN", "TN")[layout], M_, N_, K_, batch, kw.get("splitk", 1), kw.get("epi", 0), flags, s, e))
nn_ops.gemm = traced
fit_ops.gemm = traced
def full():
    opt.zero_grad(set_to_none=True)
    out = net(data["xyz"], data["cls"], chamfer_points=data["chamfer"], include_convex_loss=True, quantile=0.05,
              msc_iterations=10, max_num_clusters=25, fps_start=(data["s1"], data["s2"]))
    out[3].mean().backward(); opt.step()
for _ in range(2): full()
records.clear(); full(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for lay, M_, N_, K_, b, sk, epi, fl, s, e in records:
    key = (lay, M_, N_, K_, b, sk, epi, fl)
    agg.setdefault(key, []).append(s.elapsed_time(e) * 1e3)
tot = 0.0
rows = []
for (lay, M_, N_, K_, b, sk, epi, fl), us in agg.items():
    t = sum(us); tot += t
    fl_ = 2.0 * M_ * N_ * K_ * b
    if lay == "TN":   # A is [K, M], B is [K, N]
        byt = 4.0 * b * (K_ * M_ + K_ * N_ + M_ * N_)
    else:
        byt = 4.0 * b * (M_ * K_ + N_ * K_ + M_ * N_)
    rows.append((t, "%s M=%-7d N=%-4d K=%-7d b=%-2d sk=%-4d epi=%d %-6s x%-2d avg %7.1f us  %6.1f TF/s  %6.0f GB/s" %
                 (lay, M_, N_, K_, b, sk, epi, fl, len(us), t / len(us), fl_ / (t / len(us)) / 1e6, byt / (t / len(us)) / 1e3)))
for t, r in sorted(rows, reverse=True)[:45]:
    print("%7.0f us  %s" % (t, r))
print("total GEMM time (event-bracketed) %.2f ms in %d launches" % (tot / 1e3, len(records)))
