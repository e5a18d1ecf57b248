"""Row-sparse mean-shift backward alone (B = 24 x 2048 x 128, 10 iterations): us per call at 1 / 8 / 25 live rows per shape.
usage (GPU box): python tools/ms_rows_bench.py [lib.so ...]   -- each library in its own subprocess"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    from prifit_amd import fit_ops as F
    B, N, D, T, R = 24, 2048, 128, 10, F.KM
    gen = torch.Generator().manual_seed(0)
    proto = torch.nn.functional.normalize(torch.randn(8, D, generator=gen), dim=1)
    X = torch.nn.functional.normalize(proto[torch.randint(0, 8, (B, N), generator=gen)] + 0.1 * torch.randn(B, N, D, generator=gen), dim=2).cuda()
    bw = torch.full((B,), 0.4).cuda()
    ids = torch.stack([torch.randperm(N, generator=gen)[:R] for _ in range(B)]).cuda()
    G = torch.randn(B, R, D, generator=gen).cuda()
    with torch.no_grad():
        _, traj = F.mean_shift_trajectory(X, bw, T, keep_kernel=False)
    out = []
    for nr in [int(v) for v in os.environ.get("MS_ROWS_R", "1,8,25").split(",")]:
        nrows = torch.full((B,), nr, dtype=torch.int32).cuda()
        def run():
            Xr = X.clone().requires_grad_(True)
            c = F.MeanShiftRowsFn.apply(Xr, bw, ids, nrows, list(traj))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            c.backward(G)
            ev1.record()
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) * 1e3, Xr.grad
        for _ in range(3):
            run()
        ts = sorted(run()[0] for _ in range(20))
        out.append("R=%d %.0f us (chk %.6e)" % (nr, ts[len(ts) // 2], run()[1].double().abs().sum().item()))
    print("  ".join(out))


if __name__ == "__main__":
    if os.environ.get("MS_ROWS_ONE"):
        one()
    else:
        libs = sys.argv[1:] or [None]
        cur = os.path.join(ROOT, "prifit_amd", "lib", "libprifit_hip.so")
        keep = open(cur, "rb").read()
        try:
            for lib in libs:
                if lib and os.path.abspath(lib) != cur:
                    data = open(lib, "rb").read()
                    open(cur, "wb").write(data)
                elif lib:
                    open(cur, "wb").write(keep)
                r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MS_ROWS_ONE="1"), capture_output=True, text=True)
                print("%-40s %s" % (lib or "current", r.stdout.strip() or r.stderr.strip()[-400:]))
        finally:
            open(cur, "wb").write(keep)
