"""Timing of the streaming dW kernel (csrc/gemm_stream.hip, gemm_stream_tn_kernel) on the c3 step's shapes.
Usage: python tools/stream_tn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prifit_amd import nn_ops  # noqa: E402
from prifit_amd.nn_ops import call, ptr, cur_stream, _LL, dll  # noqa: E402

SHAPES = [(128, 96, 1572864, 128), (96, 64, 1572864, 0), (128, 64, 786432, 32), (64, 64, 786432, 0),
          (128, 128, 196608, 0), (128, 128, 49152, 0), (64, 32, 393216, 0), (32, 32, 393216, 0)]


def main():
    dev = "cuda"
    total = 0.0
    for Mo, No, P, K in SHAPES:
        g = torch.Generator(device=dev).manual_seed(1)
        G = torch.randn(P, Mo, device=dev, generator=g)
        A = torch.randn(P, No, device=dev, generator=g)
        sc = torch.rand(No, device=dev, generator=g) + 0.5
        sh = torch.randn(No, device=dev, generator=g) * 0.1
        out = torch.zeros(Mo, No, device=dev)
        ws = torch.empty(dll().prifit_gemm_stream_tn_workspace(Mo, No, _LL(P)), device=dev)
        if K:
            arg = torch.randint(0, K, (P // K, Mo), device=dev, dtype=torch.int32)
            T = torch.randn(P // K, Mo, device=dev, generator=g)
            cb = torch.randn(Mo, device=dev, generator=g) * 0.1
            cd = torch.randn(Mo, device=dev, generator=g) * 0.01

            def run():
                call("prifit_gemm_stream_tn_pool_f32", Mo, No, _LL(P), ptr(G), _LL(Mo), ptr(A), _LL(No), ptr(out), _LL(No),
                     ptr(sc), ptr(sh), ptr(arg), ptr(T), ptr(cb), ptr(cd), K, ptr(ws), cur_stream())
            onehot = torch.zeros(P // K, K, Mo, device=dev)
            onehot.scatter_(1, arg.long().unsqueeze(1), T.unsqueeze(1))
            dY = G * cb + cd + onehot.view(P, Mo)
            del onehot
        else:
            def run():
                call("prifit_gemm_stream_tn_f32", Mo, No, _LL(P), ptr(G), _LL(Mo), ptr(A), _LL(No), ptr(out), _LL(No),
                     ptr(sc), ptr(sh), ptr(ws), cur_stream())
            dY = G
        run()
        ref = dY.double().T @ torch.relu(A * sc + sh).double()
        err = ((out.double() - ref).norm() / ref.norm()).item()
        del dY, ref
        for _ in range(3):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100.0
        gb = 4.0 * P * (Mo + No) / 1e9
        tf = 2.0 * P * Mo * No / 1e12
        total += us
        print("tn[%3dx%3dx%8d pool=%3d] %8.1f us  %6.0f GB/s  %5.1f TFLOP/s  err %.1e" % (Mo, No, P, K, us, gb / us * 1e6, tf / us * 1e6, err),
              flush=True)
    print("sum %.1f us" % total)


if __name__ == "__main__":
    main()
