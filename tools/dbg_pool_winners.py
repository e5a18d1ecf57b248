"""How the winners of the pooled set-abstraction layers are distributed in the bench's workloads: fraction of (group, channel)
pairs with a gradient, distinct winning rows per group.  usage (GPU box): python tools/dbg_pool_winners.py c2|c3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PRIFIT_POOL_ALG"] = "1"
import torch
import bench
from prifit_amd import nn_ops
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
seen = {}
orig = nn_ops._pool_alg_bwd
def spy(P, G, K, Cout, Kin, W, bias, cb, cd, arg, Ttab, *rest):
    key = (P, Cout, Kin)
    if True:                                  # (the LAST call of the run is what stays)
        nz = (Ttab != 0)
        a = arg.long()
        hit = torch.zeros(G, K, dtype=torch.bool, device=arg.device)
        hit.scatter_(1, torch.where(nz, a, torch.zeros_like(a)), nz)
        rows_hit = hit.sum(1).float()
        seen[key] = (nz.float().mean().item(), rows_hit.mean().item(), K, (a == 0).float().mean().item(), Ttab.abs().max().item())
    return orig(P, G, K, Cout, Kin, W, bias, cb, cd, arg, Ttab, *rest)
nn_ops._pool_alg_bwd = spy
dev = torch.device("cuda", 0)
sys.argv = ["bench.py", "--workload", w, "--no-cpu-baseline", "--no-extra", "--steps", "3", "--warmup", "2"] + sys.argv[2:]
try:
    bench.main()
except SystemExit:
    pass
for k, v in seen.items():
    print(w, k, "T != 0: %.3f of (group, channel); rows with a winner per group: %.1f of %d; winners at row 0: %.3f; max |T| %.3e" % v)
