"""Stand-alone timing of the symmetric chord-distance kernel (2 - 2 X X^T, [24, 2048, 128]) and of the kNN selections of the
DGCNN graphs (A/B: another build of the library through PRIFIT_LIB; the product library is never overwritten)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prifit_amd import fit_ops
from prifit_amd.src import dgcnn as D

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n

g = torch.Generator(device="cuda").manual_seed(1)
X = torch.nn.functional.normalize(torch.randn(24, 2048, 128, device="cuda", generator=g), dim=2)
print("chord_sym [24 x 2048 x 128]: %.1f us" % timeit(lambda: fit_ops.chord_matrix(X, X)))
keys = []
print("chord_sym + owner keys:      %.1f us" % timeit(lambda: fit_ops.chord_matrix(X, X, [])))
pts = torch.rand(24, 2048, 3, device="cuda", generator=g) * 2 - 1
for fused in (True, False):
    D._KNN3_FUSED = fused
    print("knn graph 1 (C = 3, k = 20) fused=%s: %.1f us" % (fused, timeit(lambda: D._knn_cl(pts, 20))))
