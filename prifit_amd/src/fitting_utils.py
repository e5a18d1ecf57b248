"""Call surface of the reference's src/fitting_utils.py: the SVD with the custom backward (CustomSVD :108-139,
compute_grad_V :67-79, svd_grad_K :82-105).

On the hot path the 3x3 decomposition and this backward run inside `ellipsoid_fit_fwd/bwd` (csrc/fit.hip: Jacobi SVD
in one lane, fp64).  `customsvd` here serves callers that use the function on its own (fitting.py:5); it runs with
torch ops on whatever device the input lives on.  Importing this module does NOT reseed the global RNGs (upstream
:9-10 does)."""
import torch


def svd_grad_K(S):
    """K_ij = 1 / (sign(s_i - s_j) * max(|s_i - s_j|, 1e-6)) * 1 / (s_i + s_j) for i != j, 0 on the diagonal
    (~ 1 / (s_i^2 - s_j^2) with the gap guarded; note sign(0) = 0, so exactly equal singular values give +-inf --
    the covariance noise of src/ellipsoid_fitting.py:37-38 is what prevents that upstream)."""
    n = S.shape[0]
    col, row = S.reshape(n, 1), S.reshape(1, n)
    gap = col - row
    guarded = torch.sign(gap) * gap.abs().clamp(min=1e-6)
    eye = torch.eye(n, dtype=S.dtype, device=S.device)
    guarded = guarded * (1 - eye) + 1e-6 * eye           # the diagonal is overwritten before the inversion
    return (1 - eye) / guarded / (col + row)


def compute_grad_V(U, S, V, grad_V, grad_S):
    """dL/dM for M = U diag(S) V^T with dL/dU taken as zero:
    U diag(gS) V^T + 2 U diag(S) sym(K^T o (V^T gV)) V^T."""
    inner = svd_grad_K(S).t() * (V.t() @ grad_V)
    inner = 0.5 * (inner + inner.t())
    return U @ torch.diag(grad_S) @ V.t() + 2 * U @ torch.diag(S) @ inner @ V.t()


class CustomSVD(torch.autograd.Function):
    """forward: thin SVD (U, S, V with M = U diag(S) V^T); backward: compute_grad_V."""

    @staticmethod
    def forward(ctx, input):
        U, S, Vh = torch.linalg.svd(input, full_matrices=False)
        V = Vh.transpose(-2, -1).contiguous()
        ctx.save_for_backward(U, S, V)
        return U, S, V

    @staticmethod
    def backward(ctx, grad_U, grad_S, grad_V):
        U, S, V = ctx.saved_tensors
        return compute_grad_V(U, S, V, grad_V, grad_S)


customsvd = CustomSVD.apply
