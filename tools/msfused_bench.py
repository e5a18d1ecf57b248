"""Micro-benchmark of the fused mean-shift kernels at B=24, N=2048, D=128 (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from prifit_amd.nn_ops import call, ptr, cur_stream, _LL
B, N, D = int(os.environ.get("MSB", "24")), 2048, 128
X = torch.nn.functional.normalize(torch.randn(B, N, D, device="cuda"), dim=2)
Z = X.clone()
bw = torch.full((B,), 0.6, device="cuda")
KT = torch.empty(B, N, N, device="cuda"); Zn = torch.empty_like(Z); O = torch.empty_like(Z)
rs = torch.empty(B, N, device="cuda"); nrm = torch.empty(B, N, device="cuda")
gS = torch.empty(B, N, N, device="cuda"); gO = torch.randn(B, N, D, device="cuda"); grs = torch.randn(B, N, device="cuda"); gZ = torch.empty_like(Z)
def fwd():
    call("prifit_meanshift_fused_fwd", ptr(Z), ptr(X), ptr(bw), B, N, D, ptr(KT), _LL(N), _LL(N * N), ptr(Zn), ptr(O), ptr(rs), ptr(nrm), cur_stream())
def dz():
    call("prifit_meanshift_fused_bwd_dz", ptr(gO), _LL(N * D), ptr(X), ptr(bw), ptr(grs), ptr(KT), _LL(N), _LL(N * N), ptr(gS), B, N, D, ptr(gZ), int(os.environ.get("MSBAL", "1")), cur_stream())
def fwd_nokt():
    call("prifit_meanshift_fused_fwd", ptr(Z), ptr(X), ptr(bw), B, N, D, None, _LL(N), _LL(N * N), ptr(Zn), ptr(O), ptr(rs), ptr(nrm), cur_stream())
for name, fn in (("fused fwd", fwd), ("fwd, no K^T stream", fwd_nokt), ("fused dz", dz)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 20
    print("%-20s %7.1f us  %6.1f TF/s" % (name, us, 4.0 * B * N * N * D / us / 1e6))
print("checksum", float(Zn.sum()), float(gZ.sum()))
