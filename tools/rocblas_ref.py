"""Calibration (GPU box): what the vendor library (rocBLAS / hipBLASLt through torch.matmul, fp32) reaches on the six SA2
wide-layer products of a c3 step -- plain products, no prologue / epilogue -- next to this repo's fused kernels' rows in the
shapes table.  Not part of the product path."""
import sys, torch
dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = False
PEAK = 157.3e12
shapes = [("nt", 393216, 196, 128), ("nt", 393216, 256, 196), ("nn", 393216, 196, 256), ("nn", 393216, 128, 196),
          ("tn", 256, 196, 393216), ("tn", 196, 128, 393216), ("nt", 196608, 256, 128), ("nn", 196608, 128, 256),
          ("tn", 256, 128, 196608), ("nt", 49152, 512, 256), ("nt", 1572864, 128, 96), ("nn", 1572864, 96, 128)]
for lay, M, N, K in shapes:
    if lay == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        f = lambda: A @ B.t()
    elif lay == "nn":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        f = lambda: A @ B
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        f = lambda: A.t() @ B
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100.0
    print("%s [%d x %d x %d]  %8.1f us  %6.1f TFLOP/s  frac %.3f" % (lay, M, N, K, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / (us * 1e-6) / PEAK), flush=True)
