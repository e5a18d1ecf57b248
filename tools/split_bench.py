"""Labelled experiment: the mean-shift forward's two products on the 16-bit matrix pipe (csrc/meanshift_split.hip) against the
fp32 MFMA kernel, B = 24, N = 2048, D = 128 -- time per update (HIP events, 20 launches) and the error of ten updates
against fp64.  usage (GPU box): python tools/split_bench.py > gpurun_out/split.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prifit_amd import fit_ops as F, synth   # noqa: E402
from prifit_amd._lib import call, cur_stream, dll, ptr   # noqa: E402
import ctypes   # noqa: E402

B, N, D = 24, 2048, 128
dev = "cuda"
lab = synth.part_labels(synth.cloud("cube", B, N, 0), 8, 0)
X = torch.from_numpy(synth.prototype_embedding(lab, D, 2, noise=0.03)).to(dev)
bw = F.compute_bandwidth(X, 0.05)

X64, Z64 = X.double(), X.double()
for _ in range(10):
    S = Z64 @ X64.transpose(1, 2)
    K = torch.exp(torch.clamp((S - 1.0) / (bw.double() ** 2)[:, None, None], -13.0, 75.0))
    new = Z64 + (K @ X64 / K.sum(-1, keepdim=True) - Z64)
    Z64 = new / new.norm(dim=-1, keepdim=True)
    del S, K

out = {"shape": [B, N, D], "flops_per_update": 4.0 * B * N * N * D, "modes": {}}
O = torch.empty(B, N, D, device=dev)
rs = torch.empty(B, N, device=dev)
Zn = torch.empty_like(X)
nrm = torch.empty(B, N, device=dev)
for name in ("0", "bf16x3", "bf16x6", "fp16x3"):
    F.MS_SPLIT = name
    Z, _ = F.mean_shift_trajectory(X, bw, 10, False)
    err = ((Z.double() - Z64).abs().max() / Z64.abs().max()).item()
    mode = F.split_mode(N, D)
    if mode:
        cut = torch.empty(dll().prifit_meanshift_split_workspace(B, N, D, mode), dtype=torch.uint8, device=dev)
        call("prifit_meanshift_split_prep", ptr(X), B, N, D, mode, ptr(cut), cur_stream())

    def one():
        if mode:
            call("prifit_meanshift_split_fwd", ptr(X), ptr(cut), ptr(bw), B, N, D, mode, ptr(O), ptr(rs), cur_stream())
        else:
            call("prifit_meanshift_fused_fwd", ptr(X), ptr(X), ptr(bw), B, N, D, None, ctypes.c_longlong(N),
                 ctypes.c_longlong(N * N), ptr(Zn), ptr(O), ptr(rs), ptr(nrm), cur_stream())

    for _ in range(3):
        one()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        one()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    prep_us = None
    if mode:
        e0.record()
        for _ in range(20):
            call("prifit_meanshift_split_prep", ptr(X), B, N, D, mode, ptr(cut), cur_stream())
        e1.record()
        torch.cuda.synchronize()
        prep_us = e0.elapsed_time(e1) * 1e3 / 20
    out["modes"]["fp32" if name == "0" else name] = {
        "us_per_update": us, "prep_us_per_call": prep_us, "fp32_equivalent_TFLOPs": out["flops_per_update"] / us / 1e6,
        "err_10_updates_vs_fp64": err}
print(json.dumps(out, indent=1))
