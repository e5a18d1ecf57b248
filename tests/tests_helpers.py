import numpy as np
import torch

from prifit_amd import synth


def fit_inputs(B, N, D, seed, M=5000, noise=0.03):
    """Blob cloud (M target points, model input = fixed N-subset) + prototype embedding (oracle/make_golden_fit.py)."""
    cham, lab = synth.blobs_with_labels(B, M, seed)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    return (torch.from_numpy(cham[:, sel]), torch.from_numpy(cham),
            torch.from_numpy(synth.prototype_embedding(lab[:, sel], D, seed + 2, noise=noise)))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def intersection_case(g, dev=None, grad=True):
    """Parameters (leaf tensors), ragged surface points and dense points of fit_intersections.npz."""
    nb = len(g["Ks"])
    mv = (lambda a: _t(a).to(dev)) if dev else _t
    P = [[tuple(mv(g[f"{n}_{b}"][k]).clone().requires_grad_(grad) for n in ("r", "V", "c")) for k in range(int(g["Ks"][b]))]
         for b in range(nb)]
    return P, [mv(g[f"surf_{b}"]) for b in range(nb)], mv(g["pts"])


def check_intersection_grads(g, name, P, rtol=1e-4):
    for b in range(len(P)):
        for i, tag in enumerate("rVc"):
            want = _t(g[f"{name}_d{tag}_{b}"])
            got = torch.stack([torch.zeros_like(p[i]) if p[i].grad is None else p[i].grad for p in P[b]]).cpu()
            torch.testing.assert_close(got, want, rtol=rtol, atol=2e-6 * max(1.0, want.abs().max().item()))


def many_cluster_inputs(B=2, N=2048, D=128, seed=61, K=40, M=5000):
    """oracle/make_golden_fit.py:many_cluster_inputs -- a cloud of 40 tight blobs with 40 embedding prototypes."""
    cham, lab = synth.blobs_with_labels(B, M, seed, K=K, sigma=0.05)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    emb = synth.prototype_embedding(lab[:, sel], D, seed + 2, K=K, noise=0.03)
    return torch.from_numpy(cham[:, sel]), torch.from_numpy(cham), torch.from_numpy(emb)
