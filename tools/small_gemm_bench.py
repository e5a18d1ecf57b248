"""Stand-alone timing of the small / mid-size tiled GEMMs of the step (SA3, feature propagation, head: M = 3072 .. 49152)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from prifit_amd import nn_ops
NT, NN, TN = 0, 1, 2
CASES = [(NT, 12288, 256, 576), (NT, 3072, 256, 1536), (NT, 3072, 1024, 512), (NT, 12288, 128, 324), (NT, 49152, 128, 152),
         (NT, 12288, 128, 256), (NT, 3072, 512, 256), (NT, 3072, 256, 256), (NT, 3072, 256, 516), (NT, 49152, 50, 128),
         (NN, 3072, 512, 1024), (NN, 12288, 576, 256), (NN, 3072, 1536, 256), (NN, 12288, 324, 128), (NN, 49152, 152, 128),
         (NN, 12288, 256, 128), (NN, 3072, 256, 512), (NN, 3072, 516, 256), (NN, 3072, 256, 256),
         (TN, 256, 576, 12288), (TN, 1024, 512, 3072), (TN, 256, 1536, 3072), (TN, 128, 324, 12288), (TN, 128, 152, 49152),
         (TN, 256, 516, 3072), (TN, 128, 256, 12288), (TN, 512, 256, 3072), (TN, 256, 256, 3072)]
tot = 0.0
for lay, M, N, K in CASES:
    dev = "cuda"
    A = torch.randn((K, M) if lay == TN else (M, K), device=dev)
    B = torch.randn((N, K) if lay == NT else (K, N), device=dev)
    sk = 1
    if lay == TN:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        sk = nn_ops._splitk_for(K, tiles)
    C = torch.zeros(M, N, device=dev)
    def go():
        nn_ops.gemm(lay, M, N, K, A, A.stride(0), B, B.stride(0), C, N, splitk=sk, accumulate=sk > 1)
    for _ in range(3): go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): go()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 20
    tot += us
    print("%s [%6d x %4d x %5d] sk %3d  %7.1f us  %6.1f TF/s" % (("NT", "NN", "TN")[lay], M, N, K, sk, us, 2.0 * M * N * K / us / 1e6))
print("total %.1f us" % tot)
