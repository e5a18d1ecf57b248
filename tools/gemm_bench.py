"""Micro-benchmark of prifit_gemm_f32 on the shapes of the hot path (run on the GPU box).
usage: python tools/gemm_bench.py [filter]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from prifit_amd import nn_ops

NT, NN, TN = 0, 1, 2


def run(name, lay, M, N, K, batch=1, epi=0, affine=False, stats=False, splitk=1, iters=20):
    dev = "cuda"
    A = torch.randn(batch, (K if lay == TN else M), (M if lay == TN else K), device=dev)
    B = torch.randn(batch, (N if lay == NT else K), (K if lay == NT else N), device=dev)
    C = torch.zeros(batch, M, N, device=dev)
    aff = (torch.rand(4096, device=dev), torch.rand(4096, device=dev)) if affine else None
    bw = torch.full((batch,), 0.5, device=dev)
    aux = torch.rand(batch, M, N, device=dev) if epi == 3 else None
    slab = torch.empty(max((M + 127) // 128, 4096), 2, N, device=dev) if stats else None
    kw = dict(batch=batch, sA=A.stride(0), sB=B.stride(0), sC=M * N, splitk=splitk, epi=epi, accumulate=splitk > 1,
              epi_scalar=bw if epi >= 2 else None, aux=aux, ld_aux=N, s_aux=M * N, stats=slab)
    if affine:
        kw["b_affine" if lay == TN else "a_affine"] = aff

    def go():
        nn_ops.gemm(lay, M, N, K, A, A.stride(1), B, B.stride(1), C, N, **kw)

    for _ in range(3):
        go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        go()
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    fl = 2.0 * M * N * K * batch
    by = 4.0 * batch * (M * K + N * K + M * N)
    print("%-34s %9.1f us  %6.1f TF/s  %6.0f GB/s(alg)" % (name, us, fl / us / 1e6, by / us / 1e3), flush=True)


CASES = [
    ("ms S=ZX^T NT 2048^2x128 b24 mskern", NT, 2048, 2048, 128, dict(batch=24, epi=2)),
    ("ms S plain NT 2048^2x128 b24", NT, 2048, 2048, 128, dict(batch=24)),
    ("ms S NOSTORE NT 2048^2x128 b24", NT, 2048, 2048, 128, dict(batch=24, epi=100)),
    ("big square NOSTORE NT 4096^3", NT, 4096, 4096, 4096, dict(epi=100)),
    ("ms O=KX NN 2048x128x2048 b24", NN, 2048, 128, 2048, dict(batch=24)),
    ("ms skinny NN sk1 NOSTORE", NN, 2048, 128, 2048, dict(batch=24, splitk=1, epi=100)),
    ("ms skinny NN sk4 NOSTORE", NN, 2048, 128, 2048, dict(batch=24, splitk=4, epi=100)),
    ("ms skinny NT sk1", NT, 2048, 128, 2048, dict(batch=24, splitk=1)),
    ("ms skinny NT sk4", NT, 2048, 128, 2048, dict(batch=24, splitk=4)),
    ("ms skinny NT sk4 NOSTORE", NT, 2048, 128, 2048, dict(batch=24, splitk=4, epi=100)),
    ("ms skinny NN sk2", NN, 2048, 128, 2048, dict(batch=24, splitk=2)),
    ("ms skinny NN sk4", NN, 2048, 128, 2048, dict(batch=24, splitk=4)),
    ("ms skinny NN sk8", NN, 2048, 128, 2048, dict(batch=24, splitk=8)),
    ("ms skinny TN sk1", TN, 2048, 128, 2048, dict(batch=24, splitk=1)),
    ("ms skinny TN sk2", TN, 2048, 128, 2048, dict(batch=24, splitk=2)),
    ("ms skinny TN sk4", TN, 2048, 128, 2048, dict(batch=24, splitk=4)),
    ("ms skinny TN sk8", TN, 2048, 128, 2048, dict(batch=24, splitk=8)),
    ("ms gS NT 2048^2x128 b24 msbwd", NT, 2048, 2048, 128, dict(batch=24, epi=3)),
    ("ms dX TN 2048x128x2048 b24", TN, 2048, 128, 2048, dict(batch=24)),
    ("sa1.3 L3 fwd NT P=1.57M 96->128", NT, 1572864, 128, 96, dict(affine=True, stats=True)),
    ("sa1.3 L2 fwd NT P=1.57M 64->96", NT, 1572864, 96, 64, dict(affine=True, stats=True)),
    ("sa1.3 L1 fwd NT P=1.57M 8->64", NT, 1572864, 64, 8, dict(stats=True)),
    ("sa1.3 L3 dA NN P=1.57M 128->96", NN, 1572864, 96, 128, {}),
    ("sa1.3 L2 dA NN P=1.57M 96->64", NN, 1572864, 64, 96, {}),
    ("sa1.2 L2 fwd NT P=786K 64->64", NT, 786432, 64, 64, dict(affine=True, stats=True)),
    ("sa1.2 L3 fwd NT P=786K 64->128", NT, 786432, 128, 64, dict(affine=True, stats=True)),
    ("sa1.2 L3 dA NN P=786K 128->64", NN, 786432, 64, 128, {}),
    ("sa2.1 L2 fwd NT P=196K 128->128", NT, 196608, 128, 128, dict(affine=True, stats=True)),
    ("sa1.3 L3 dW TN 128x96 P=1.57M", TN, 128, 96, 1572864, dict(affine=True, splitk=512)),
    ("sa2.2 L1 fwd NT P=393K 324->128", NT, 393216, 128, 324, dict(stats=True)),
    ("sa2.2 L3 fwd NT P=393K 196->256", NT, 393216, 256, 196, dict(affine=True, stats=True)),
    ("sa2.2 L1 dW TN 128x324 P=393K", TN, 128, 324, 393216, dict(splitk=128)),
    ("sa3 L3 fwd NT P=3072 512->1024", NT, 3072, 1024, 512, dict(affine=True, stats=True)),
    ("fp1 L1 fwd NT P=49K 152->128", NT, 49152, 128, 152, dict(stats=True)),
    ("big square NT 4096^3", NT, 4096, 4096, 4096, {}),
]

def run_dw(name, Mo, No, P, iters=20):
    dev = "cuda"
    dY, A = torch.randn(P, Mo, device=dev), torch.randn(P, No, device=dev)
    aff = (torch.rand(No, device=dev), torch.rand(No, device=dev))
    out = torch.zeros(Mo, No, device=dev)
    go = lambda: nn_ops._weight_grad(dY, P, Mo, A, No, aff, out=out)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        go()
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    print("%-34s %9.1f us  %6.1f TF/s  %6.0f GB/s(alg)" % (name, us, 2.0 * Mo * No * P / us / 1e6, 4.0 * P * (Mo + No) / us / 1e3), flush=True)


DW_CASES = [("sa1.3 dW 128x96 P=1.57M", 128, 96, 1572864), ("sa1.3 dW 96x64 P=1.57M", 96, 64, 1572864),
            ("sa1.2 dW 128x64 P=786K", 128, 64, 786432), ("sa1.2 dW 64x64 P=786K", 64, 64, 786432),
            ("sa1.1 dW 64x32 P=393K", 64, 32, 393216), ("sa1.1 dW 32x32 P=393K", 32, 32, 393216),
            ("sa2.1 dW 128x128 P=196K", 128, 128, 196608)]

if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, lay, M, N, K, kw in CASES:
        if flt in name:
            run(name, lay, M, N, K, **kw)
    for name, Mo, No, P in DW_CASES:
        if flt in name:
            run_dw(name, Mo, No, P)
