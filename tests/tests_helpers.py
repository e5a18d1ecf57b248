import numpy as np
import torch

from prifit_amd import synth


def fit_inputs(B, N, D, seed, M=5000, noise=0.03):
    """Blob cloud (M target points, model input = fixed N-subset) + prototype embedding (oracle/make_golden_fit.py)."""
    cham, lab = synth.blobs_with_labels(B, M, seed)
    sel = np.random.default_rng(seed + 1).choice(M, N, replace=False)
    return (torch.from_numpy(cham[:, sel]), torch.from_numpy(cham),
            torch.from_numpy(synth.prototype_embedding(lab[:, sel], D, seed + 2, noise=noise)))
