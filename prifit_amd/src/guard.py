"""Call surface of the reference's src/guard.py (guarded elementwise routines).  On the hot path these clamps live
inside the kernels (meanshift_fused.hip, meanshift.hip, fit.hip); the functions are kept for callers of the module."""
import torch


def guard_exp(x, max_value=75, min_value=-13):
    """upstream :6-11"""
    return torch.exp(torch.clamp(x, max=max_value, min=min_value))


def guard_sqrt(x, minimum=1e-5):
    """upstream :13-18"""
    return torch.sqrt(torch.clamp(x, min=minimum))


def guard_acos(x):
    """upstream :21-23"""
    return torch.acos(torch.clamp(x, min=-1.0, max=1.0))
