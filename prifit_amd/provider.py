"""Batch augmentation under the reference's names (provider.py:265-303), the ones its trainer calls
(train_partseg_shapenet.py:372-373):

    points[:, :, 0:3] = provider.random_scale_point_cloud(points[:, :, 0:3])
    points[:, :, 0:3] = provider.shift_point_cloud(points[:, :, 0:3])

A numpy batch is handled the reference's way -- one `np.random.uniform` draw per call, in the reference's order, applied in
place and returned -- so a seeded run reproduces the reference's batches bit for bit (tests/golden/data_readers.npz).  A torch
tensor (any device) gets the same law from torch's generator in one batched op on ITS device: that is the form the trainer of
this package uses (`train_step.random_scale_shift`), no host round trip per batch.
"""
import numpy as np
import torch


def _is_np(x):
    return isinstance(x, np.ndarray)


def random_scale_point_cloud(batch_data, scale_low=0.8, scale_high=1.25, generator=None):
    """One scale per cloud, uniform in [scale_low, scale_high] (provider.py:292-303).  [B,N,3] in, the same object out."""
    B = batch_data.shape[0]
    if _is_np(batch_data):
        scales = np.random.uniform(scale_low, scale_high, B)
        # (float64 draws against a float32 batch: the product is formed in double and rounded once, as the reference's
        # per-cloud `batch_data[b] *= scales[b]` does under this numpy's promotion rules)
        np.multiply(batch_data, scales.reshape(B, 1, 1), out=batch_data, casting="same_kind")
        return batch_data
    scales = torch.empty(B, 1, 1, device=batch_data.device, dtype=batch_data.dtype).uniform_(scale_low, scale_high, generator=generator)
    return batch_data.mul_(scales)


def shift_point_cloud(batch_data, shift_range=0.1, generator=None):
    """One shift per cloud, uniform in [-shift_range, shift_range]^3 (provider.py:278-289)."""
    B = batch_data.shape[0]
    if _is_np(batch_data):
        shifts = np.random.uniform(-shift_range, shift_range, (B, 3))
        np.add(batch_data, shifts.reshape(B, 1, 3), out=batch_data, casting="same_kind")
        return batch_data
    shifts = torch.empty(B, 1, 3, device=batch_data.device, dtype=batch_data.dtype).uniform_(-shift_range, shift_range, generator=generator)
    return batch_data.add_(shifts)


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05, generator=None):
    """Per-point Gaussian jitter clipped to [-clip, clip] (provider.py:265-276); returns a new batch."""
    assert clip > 0
    if _is_np(batch_data):
        B, N, C = batch_data.shape
        jittered = np.clip(sigma * np.random.randn(B, N, C), -1 * clip, clip)
        jittered += batch_data
        return jittered
    noise = torch.randn(batch_data.shape, device=batch_data.device, dtype=batch_data.dtype, generator=generator)
    return batch_data + (sigma * noise).clamp_(-clip, clip)
