"""Phase timing of the fused mean-shift forward kernel from in-kernel s_memtime stamps (DIAGNOSIS build only).

    touch prifit_amd/csrc/meanshift_fused.hip; PRIFIT_BUILD_DEFS=-DMSF_STAMPS python -m prifit_amd.build
    python tools/msf_stamps.py        (GPU box; rebuild without the flag afterwards)

Wave 0 of eight workgroups stamps eight phase boundaries in each of its first 16 key steps:
0 loop top | 1 after barrier 1 | 2 tile staged (vmcnt wait + 8 ds_write) | 3 after barrier 2 | 4 stream stores + next tile
requested | 5 S product done | 6 transform done | 7 O product done."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from prifit_amd.nn_ops import call, ptr, cur_stream, _LL, dll

B, N, D = 24, 2048, 128
X = torch.nn.functional.normalize(torch.randn(B, N, D, device="cuda"), dim=2)
Z = X.clone()
bw = torch.full((B,), 0.6, device="cuda")
KT = torch.empty(B, N, N, device="cuda"); Zn = torch.empty_like(Z); O = torch.empty_like(Z)
rs = torch.empty(B, N, device="cuda"); nrm = torch.empty(B, N, device="cuda")
for _ in range(5):
    call("prifit_meanshift_fused_fwd", ptr(Z), ptr(X), ptr(bw), B, N, D, ptr(KT), _LL(N), _LL(N * N), ptr(Zn), ptr(O), ptr(rs), ptr(nrm), cur_stream())
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16 * 8))()
assert dll().prifit_debug_msf_stamps(buf, 8 * 16 * 8) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 16, 8).astype(np.int64)
names = ["barrier 1", "stage tile (vmcnt + ds_write)", "barrier 2", "stores + next tile request", "S product", "transform", "O product"]
print("workgroup slot: cycles per phase, median over key steps 2..15 (s_memtime ticks = shader cycles)")
for slot in range(8):
    d = np.diff(t[slot], axis=1)[2:]                       # [steps, 7]
    step = (t[slot, 3:, 0] - t[slot, 2:-1, 0])
    gap = t[slot, 3:, 0] - t[slot, 2:-1, 7]                 # O done -> next loop top
    print("slot %d  start %d" % (slot, t[slot, 0, 0] - t[:, 0, 0].min()), " step %6d" % np.median(step), " | ".join("%s %5d" % (n, v) for n, v in zip(names, np.median(d, axis=0))), "| loop back %d" % np.median(gap))
