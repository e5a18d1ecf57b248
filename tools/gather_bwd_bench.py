"""Stand-alone timing of the first-layer backward of a gather-mode set-abstraction level (SA2 shapes of the MSG network):
bn_relu_bwd_apply + prifit_gather_linear_bwd (global atomics) against prifit_gather_linear_bwd_bn (fused, LDS-staged)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prifit_amd._lib import call, cur_stream, ptr
LL = ctypes.c_longlong
F = ctypes.c_float
B, N, S, C = 24, 512, 128, 128
dev = "cuda"
for K in (64, 128):
    P = B * S * K
    g = torch.Generator(device=dev).manual_seed(K)
    G = torch.randn(P, C, device=dev, generator=g)
    Y = torch.randn(P, C, device=dev, generator=g)
    idx = torch.randint(0, N, (B, S, K), device=dev, generator=g, dtype=torch.int32)
    if os.environ.get("PAD", "0") == "1":   # sparse cloud: ~60 % of the slots repeat the first index
        cnt = torch.randint(1, K, (B, S, 1), device=dev, generator=g)
        idx = torch.where(torch.arange(K, device=dev).view(1, 1, K) < cnt, idx, idx[:, :, :1])
    vec = [torch.randn(C, device=dev, generator=g) for _ in range(5)]
    dY = torch.empty(P, C, device=dev)
    def old():
        dU = torch.zeros(B, N, C, device=dev); dV = torch.empty(B, S, C, device=dev)
        call("prifit_bn_relu_bwd_apply", ptr(G), LL(C), ptr(Y), LL(C), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(vec[4]), P, C, 0, F(0.0), ptr(dY), LL(C), cur_stream())
        call("prifit_gather_linear_bwd", ptr(dY), ptr(idx), B, N, S, K, C, ptr(dU), ptr(dV), cur_stream())
        return dU, dV
    def new():
        dU = torch.zeros(B, N, C, device=dev); dV = torch.zeros(B, S, C, device=dev)
        call("prifit_gather_linear_bwd_bn", ptr(G), ptr(Y), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), ptr(vec[4]), ptr(idx), B, N, S, K, C, ptr(dU), ptr(dV), cur_stream())
        return dU, dV
    a, b = old(), new()
    print("K=%d max|dU diff| %.2e (|dU| %.1f)  max|dVc diff| %.2e" % (K, (a[0] - b[0]).abs().max().item(), a[0].abs().max().item(), (a[1] - b[1]).abs().max().item()))
    for name, fn in (("apply + atomics", old), ("fused LDS", new)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("  %-16s %.1f us" % (name, 1e3 * e0.elapsed_time(e1) / 20))
