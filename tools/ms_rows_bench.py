"""Row-sparse mean-shift backward alone (B = 24 x 2048 x 128, 10 iterations): us per call of prifit_meanshift_rows_bwd in its
three modes (2: a workgroup per live row, 1: key-tiled iterations over a work queue, 0: one launch per iteration) at a given
number of live rows per shape, and the largest difference between modes 2 and 0.  Another build of the library: PRIFIT_LIB.
usage (GPU box): python tools/ms_rows_bench.py [live rows per shape, default 8]
                 rocprofv3 --kernel-trace --stats --output-format csv -d out -o p -- python3 tools/ms_rows_bench.py 8   (per-kernel averages)
(profiles/r06_ms_rows.txt)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prifit_amd import fit_ops as F
from prifit_amd._lib import call, cur_stream, ptr, query
B, N, D, T = 24, 2048, 128, 10
live = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = 32 if live <= 32 else 64
g = torch.Generator().manual_seed(0)
proto = torch.nn.functional.normalize(torch.randn(8, D, generator=g), dim=1)
X = torch.nn.functional.normalize(proto[torch.randint(0, 8, (B, N), generator=g)] + 0.1 * torch.randn(B, N, D, generator=g), dim=2).cuda()
bw = torch.full((B,), 0.4).cuda()
with torch.no_grad():
    Zf, traj = F.mean_shift_trajectory(X, bw, T, keep_kernel=False)
ids = torch.stack([torch.randperm(N, generator=g)[:R] for _ in range(B)]).cuda()
nrows = torch.full((B,), live, dtype=torch.int32).cuda()
G = torch.randn(B, R, D, generator=g).cuda()
ws = torch.empty(query("prifit_meanshift_rows_bwd_workspace", B, N, D, R, T), dtype=torch.float32, device="cuda")
arr = lambda k: (ctypes.c_void_p * T)(*[it[k].data_ptr() for it in traj])
res = {}
for mode in (2, 1, 0):
    gX = torch.zeros(B, N, D, device="cuda")
    def run():
        call("prifit_meanshift_rows_bwd", ptr(X), ptr(bw), B, N, D, T, arr(0), arr(4), arr(2), arr(3), arr(5), ptr(ids), ptr(nrows), R, ptr(G), ptr(ws), ptr(gX), mode, cur_stream())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    gX.zero_(); run(); torch.cuda.synchronize()
    res[mode] = (e0.elapsed_time(e1) / 50 * 1e3, gX.clone())
print(os.environ.get("PRIFIT_LIB", "default").split("/")[-1], "live", live, " ".join("mode%d %.1f us" % (m, res[m][0]) for m in res),
      "max|2-0| %.2e of %.2e" % ((res[2][1] - res[0][1]).abs().max().item(), res[0][1].abs().max().item()))
