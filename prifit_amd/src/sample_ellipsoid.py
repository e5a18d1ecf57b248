"""Call surface of the reference's src/sample_ellipsoid.py (SampleEllipsoid :12-96, Loss :99-125).

trimesh's even surface sampler (upstream :31-35, :77-84) is absent by design: its role -- a fixed, gradient-free table
of surface parameters -- is played by the build's deterministic tables (a Fibonacci sphere for the ellipsoid,
csrc/fit.hip `fib_dir`; the same recipe here in torch so that stand-alone callers get the points themselves).
On the hot path sampling and the nearest-neighbour search are one kernel (fit_ops.SampleNNLossFn)."""
import math

import torch


def fibonacci_uv(n, device):
    """(cos U, sin U, cos V, sin V) of sample j of n: z_j = 1 - (2j+1)/n, longitude 2 pi frac(j / phi), evaluated in
    float64 and rounded to fp32 once, exactly like csrc/fit.hip fib_dir."""
    j = torch.arange(int(n), dtype=torch.float64, device=device)
    z = 1.0 - (2.0 * j + 1.0) / float(n)
    lon = 2.0 * math.pi * torch.frac(j * 0.6180339887498949)
    return (torch.cos(lon).float(), torch.sin(lon).float(), z.float(), torch.sqrt((1.0 - z * z).clamp(min=0.0)).float())


def cuboid_unit_table(n, a, b, c, device):
    """The unit-box surface parameters of csrc/fit.hip cuboid_unit for samples 0..n-1, float64 then fp32 once."""
    j = torch.arange(int(n), dtype=torch.float64, device=device)
    w = torch.tensor([a * b, a * b, b * c, b * c, c * a, c * a], dtype=torch.float64, device=device)
    edges = torch.cumsum(w, 0)[:5] / w.sum()
    t = (j + 0.5) / float(n)
    face = (t.unsqueeze(1) >= edges.unsqueeze(0)).sum(dim=1)            # the last edge not above t
    s1 = 2.0 * torch.frac(0.5 + j * 0.7548776662466927) - 1.0
    s2 = 2.0 * torch.frac(0.5 + j * 0.5698402909980532) - 1.0
    sg = torch.where(face % 2 == 1, -torch.ones_like(s1), torch.ones_like(s1))
    v = torch.where((face < 2).unsqueeze(1), torch.stack([s1, s2, sg], 1),
                    torch.where((face < 4).unsqueeze(1), torch.stack([sg, s1, s2], 1), torch.stack([s2, sg, s1], 1)))
    side = torch.tensor([a, b, c], dtype=torch.float64, device=device)
    return ((v * side) / (side + 1e-6)).float()


class SampleEllipsoid:
    def sample(self, a, b, c, center, transformation, n=500):
        """upstream :17-53: `n` points on the ellipsoid with semi-axes (a, b, c), rotated by `transformation` and moved
        to `center`; gradients flow to a, b, c, transformation and center, the (U, V) parameters are constants."""
        cu, su, cv, sv = fibonacci_uv(n, transformation.device)
        pts = torch.stack([a * cu * sv, b * su * sv, c * cv], 1)     # uniform_sample_points_on_ellipsoid, :55-63
        return pts @ transformation.T + center, None

    def sample_cuboid(self, a, b, c, center, transformation, n=500):
        """upstream :65-96: `n` points on the surface of the box with half-sides (a, b, c) (sides 2a, 2b, 2c), rotated by
        `transformation` and moved to `center`.  The unit-box parameters are constants (trimesh's sampler upstream, the
        build's table here: csrc/fit.hip cuboid_unit -- area-proportional faces [+z,-z,+x,-x,+y,-y], R2 sequence inside
        a face, scaled (u s) / (s + 1e-6) as upstream :88); gradients flow to a, b, c, transformation and center."""
        sa, sb, sc = (float(t.detach()) if torch.is_tensor(t) else float(t) for t in (a, b, c))
        u = cuboid_unit_table(int(n), sa, sb, sc, transformation.device)
        sides = torch.stack([a, b, c]).view(1, 3)
        return (u * sides) @ transformation.T + center, None

    def uniform_sample_points_on_ellipsoid(self, U, V, a, b, c):
        """upstream :55-63"""
        return torch.stack([a * torch.cos(U) * torch.sin(V), b * torch.sin(U) * torch.sin(V), c * torch.cos(V)], 1)


class Loss:
    """upstream :99-125 without the Open3D drawing: mean over shapes of the two-sided chamfer distance between the
    ground-truth cloud and the resampled surface points."""

    def loss(self, gt_points, sampled_points):
        from .utils import chamfer_distance_kdtree
        d = [chamfer_distance_kdtree(sampled_points[b].unsqueeze(0), gt_points[b].unsqueeze(0))
             for b in range(gt_points.shape[0]) if torch.is_tensor(sampled_points[b])]
        return torch.mean(torch.stack(d))
