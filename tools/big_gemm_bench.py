"""Stand-alone timing of the big tiled products of SA2 (plain forms: no statistics / BatchNorm-backward epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from prifit_amd import nn_ops
NT, NN = 0, 1
CASES = [(NN, 393216, 128, 256), (NN, 393216, 128, 196), (NT, 393216, 196, 128), (NT, 393216, 256, 196), (NN, 393216, 196, 256),
         (NN, 196608, 128, 256), (NT, 196608, 256, 128)]
for lay, M, N, K in CASES:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn((N, K) if lay == NT else (K, N), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    def go():
        nn_ops.gemm(lay, M, N, K, A, K, B, B.stride(0), C, N)
    for _ in range(3): go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): go()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 10
    print("%s [%6d x %4d x %5d]  %7.1f us  %6.1f TF/s" % (("NT", "NN")[lay], M, N, K, us, 2.0 * M * N * K / us / 1e6))
