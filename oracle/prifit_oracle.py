"""CPU oracle for the PRIFIT hot path -- TEST INFRASTRUCTURE, not product code.

A PyTorch-CPU / numpy restatement of the reference's algorithm for the path named in
BASELINE.json (PointNet++ MSG set-abstraction / feature-propagation stack + mean-shift driven
ellipsoid fitting).  Each function cites the upstream file:line it follows.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product package `prifit_amd` never does (it fails loudly without its HIP library).

Parity pinned: YES.  The reference is Python, so it is imported in the build container by
`oracle/make_golden.py` (through `oracle/refshim.py`), which (a) checks every function of this
file against the reference on seeded inputs and (b) writes the golden vectors under
`tests/golden/`.  `tests/test_oracle_golden.py` re-checks this file against those vectors
wherever the tests run.  Index-producing ops additionally have a scalar C restatement with
explicit rounding in `oracle/prifit_oracle.c` (loaded here through ctypes as `clib()`).

Layout convention of this file: "channels-last" -- clouds are [B, N, 3], features [B, N, C].
The reference's modules take channels-first tensors ([B, C, N]); the thin `Oracle*` modules at
the bottom keep that outer surface and the reference's state_dict keys.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def build_clib(force=False):
    so = os.path.join(_HERE, "libprifit_oracle.so")
    src = os.path.join(_HERE, "prifit_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libprifit_oracle.so"])
    return so


def clib():
    """ctypes handle on oracle/libprifit_oracle.so (built on demand with gcc)."""
    global _CLIB
    if _CLIB is None:
        _CLIB = ctypes.CDLL(build_clib())
    return _CLIB


def _fp(t):
    return ctypes.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------------------------
# index ops -- C restatement (bit-exact recipes) wrapped for torch CPU tensors
# ----------------------------------------------------------------------------------------------
def c_square_distance(src, dst):
    src = src.contiguous().float()
    dst = dst.contiguous().float()
    B, S, _ = src.shape
    N = dst.shape[1]
    out = torch.empty(B, S, N, dtype=torch.float32)
    clib().orc_square_distance(_fp(src), _fp(dst), B, S, N, _fp(out))
    return out


def c_farthest_point_sample(xyz, npoint, start_idx):
    xyz = xyz.contiguous().float()
    B, N, _ = xyz.shape
    start = start_idx.contiguous().to(torch.int64)
    out = torch.empty(B, npoint, dtype=torch.int64)
    clib().orc_fps(_fp(xyz), B, N, npoint, _fp(start), _fp(out))
    return out


def c_query_ball_point(radius, nsample, xyz, new_xyz):
    xyz = xyz.contiguous().float()
    new_xyz = new_xyz.contiguous().float()
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty(B, S, nsample, dtype=torch.int64)
    r2 = ctypes.c_float(float(np.float32(radius ** 2)))
    clib().orc_ball_query(_fp(xyz), _fp(new_xyz), B, N, S, r2, nsample, _fp(out))
    return out


def c_three_nn(xyz1, xyz2):
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = torch.empty(B, N, 3, dtype=torch.int64)
    d = torch.empty(B, N, 3, dtype=torch.float32)
    clib().orc_three_nn(_fp(xyz1), _fp(xyz2), B, N, S, _fp(idx), _fp(d))
    return d, idx


# ----------------------------------------------------------------------------------------------
# index ops -- torch restatement (same complexity class as the reference: matmul + full sort)
# ----------------------------------------------------------------------------------------------
def square_distance(src, dst):
    """models/pointnet_util.py:19-40 -- expanded form, accumulated in the reference's order."""
    d = torch.matmul(src, dst.transpose(1, 2)) * -2.0
    sx, sy, sz = src.unbind(-1)
    dx, dy, dz = dst.unbind(-1)
    d = d + ((sx * sx + sy * sy) + sz * sz).unsqueeze(2)
    d = d + ((dx * dx + dy * dy) + dz * dz).unsqueeze(1)
    return d


def gather_rows(table, idx):
    """models/pointnet_util.py:43-60 index_points: table [B, N, C], idx [B, ...] -> [B, ..., C]."""
    B = table.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.gather(table, 1, flat.unsqueeze(-1).expand(-1, -1, table.shape[-1]))
    return out.reshape(*idx.shape, table.shape[-1])


def farthest_point_sample(xyz, npoint, start_idx):
    """models/pointnet_util.py:63-84; `start_idx` [B] replaces the torch.randint at line 75."""
    B, N, _ = xyz.shape
    picked = torch.empty(B, npoint, dtype=torch.int64)
    nearest = torch.full((B, N), 1e10, dtype=xyz.dtype)
    cur = start_idx.to(torch.int64).clone()
    rows = torch.arange(B)
    for i in range(npoint):
        picked[:, i] = cur
        c = xyz[rows, cur].unsqueeze(1)
        diff = xyz - c
        sq = diff * diff
        d = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        nearest = torch.where(d < nearest, d, nearest)
        cur = torch.argmax(nearest, dim=1)
    return picked


def query_ball_point(radius, nsample, xyz, new_xyz):
    """models/pointnet_util.py:87-107 (sort-based, like the reference)."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d = square_distance(new_xyz, xyz)
    ids = torch.arange(N, dtype=torch.int64).expand(B, S, N)
    ids = torch.where(d > radius ** 2, torch.full_like(ids, N), ids)
    ids = ids.sort(dim=-1)[0][:, :, :nsample]
    first = ids[:, :, :1].expand(-1, -1, nsample)
    return torch.where(ids == N, first, ids)


def three_nn(xyz1, xyz2):
    """models/pointnet_util.py:291-293: three smallest expanded-form distances via a full sort."""
    d = square_distance(xyz1, xyz2)
    d, idx = d.sort(dim=-1)
    return d[:, :, :3], idx[:, :, :3]


def three_interpolate(points2, d3, idx3):
    """models/pointnet_util.py:295-298: inverse-distance weights (no clamp; d may be < 0)."""
    recip = 1.0 / (d3 + 1e-8)
    w = recip / recip.sum(dim=2, keepdim=True)
    return (gather_rows(points2, idx3) * w.unsqueeze(-1)).sum(dim=2)


# ----------------------------------------------------------------------------------------------
# shared per-position MLP: (1x1 conv + train/eval BatchNorm + ReLU) applied on channels-last rows
# ----------------------------------------------------------------------------------------------
def _pointwise_block(x2d, conv, bn):
    """x2d [P, Cin] -> [P, Cout]; conv is a Conv1d/Conv2d with kernel 1, bn its BatchNorm."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    y = F.linear(x2d, w, conv.bias)
    # BatchNorm over all P positions.  Fed as [1, C, P] so that ATen takes its channels-first path
    # (cascade summation): the [P, C] path accumulates 1e6-element columns naively in fp32 and is
    # 1e-3 off, which the reference's [B, C, K, S] layout does not suffer from.
    y = F.batch_norm(y.t().unsqueeze(0), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training, bn.momentum, bn.eps).squeeze(0).t()
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return F.relu(y)


class OracleSetAbstractionMsg(nn.Module):
    """models/pointnet_util.py:204-261 (state_dict keys: conv_blocks.i.j / bn_blocks.i.j)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for widths in mlp_list:
            convs, bns, last = nn.ModuleList(), nn.ModuleList(), in_channel + 3
            for w in widths:
                convs.append(nn.Conv2d(last, w, 1))
                bns.append(nn.BatchNorm2d(w))
                last = w
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points, fps_start=None):
        xyz = xyz.transpose(1, 2).contiguous()
        feats = points.transpose(1, 2).contiguous() if points is not None else None
        B, N, _ = xyz.shape
        S = self.npoint
        if fps_start is None:
            fps_start = torch.randint(0, N, (B,), dtype=torch.long)
        centres = gather_rows(xyz, farthest_point_sample(xyz, S, fps_start))
        pooled = []
        for radius, K, convs, bns in zip(self.radius_list, self.nsample_list, self.conv_blocks, self.bn_blocks):
            gi = query_ball_point(radius, K, xyz, centres)
            rel = gather_rows(xyz, gi) - centres.unsqueeze(2)
            g = torch.cat([gather_rows(feats, gi), rel], dim=-1) if feats is not None else rel  # :247 order
            x = g.reshape(B * S * K, -1)
            for conv, bn in zip(convs, bns):
                x = _pointwise_block(x, conv, bn)
            pooled.append(x.reshape(B, S, K, -1).max(dim=2)[0])
        out = torch.cat(pooled, dim=-1)
        return centres.transpose(1, 2).contiguous(), out.transpose(1, 2).contiguous()


class OracleSetAbstraction(nn.Module):
    """models/pointnet_util.py:160-201 (single scale, or group_all)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for w in mlp:
            self.mlp_convs.append(nn.Conv2d(last, w, 1))
            self.mlp_bns.append(nn.BatchNorm2d(w))
            last = w

    def forward(self, xyz, points, fps_start=None):
        xyz = xyz.transpose(1, 2).contiguous()
        feats = points.transpose(1, 2).contiguous() if points is not None else None
        B, N, _ = xyz.shape
        if self.group_all:
            centres = torch.zeros(B, 1, 3, dtype=xyz.dtype)
            g = torch.cat([xyz, feats], dim=-1) if feats is not None else xyz  # :154 order
            g = g.unsqueeze(1)
            S, K = 1, N
        else:
            S, K = self.npoint, self.nsample
            if fps_start is None:
                fps_start = torch.randint(0, N, (B,), dtype=torch.long)
            centres = gather_rows(xyz, farthest_point_sample(xyz, S, fps_start))
            gi = query_ball_point(self.radius, K, xyz, centres)
            rel = gather_rows(xyz, gi) - centres.unsqueeze(2)
            g = torch.cat([rel, gather_rows(feats, gi)], dim=-1) if feats is not None else rel  # :131 order
        x = g.reshape(B * S * K, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = _pointwise_block(x, conv, bn)
        out = x.reshape(B, S, K, -1).max(dim=2)[0]
        return centres.transpose(1, 2).contiguous(), out.transpose(1, 2).contiguous()


class OracleFeaturePropagation(nn.Module):
    """models/pointnet_util.py:264-314."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for w in mlp:
            self.mlp_convs.append(nn.Conv1d(last, w, 1))
            self.mlp_bns.append(nn.BatchNorm1d(w))
            last = w

    def forward(self, xyz1, xyz2, points1, points2):
        xyz1 = xyz1.transpose(1, 2)
        xyz2 = xyz2.transpose(1, 2)
        p2 = points2.transpose(1, 2)
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interp = p2.expand(-1, N, -1)
        else:
            d3, i3 = three_nn(xyz1, xyz2)
            interp = three_interpolate(p2, d3, i3)
        x = torch.cat([points1.transpose(1, 2), interp], dim=-1) if points1 is not None else interp
        x = x.reshape(B * N, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = _pointwise_block(x, conv, bn)
        return x.reshape(B, N, -1).transpose(1, 2).contiguous()


class OracleMSGPartSeg(nn.Module):
    """models/pointnet2_part_seg_msg.py:11-134 / models/pretrain_pointnet2_part_seg_msg.py:11-88.

    Returns the trainer's 5-tuple (train_partseg_shapenet.py:387) and, when the convex loss is on,
    labels / ellipse params / embedding as elements 6-8 (pointnet2_part_seg_msg.py:134)."""

    def __init__(self, num_parts, normal_channel=False):
        super().__init__()
        extra = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.beta = 1
        self.sa1 = OracleSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], 3 + extra,
                                           [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = OracleSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 128 + 128 + 64,
                                           [[128, 128, 256], [128, 196, 256]])
        self.sa3 = OracleSetAbstraction(None, None, None, 512 + 3, [256, 512, 1024], True)
        self.fp3 = OracleFeaturePropagation(1536, [256, 256])
        self.fp2 = OracleFeaturePropagation(576, [256, 128])
        self.fp1 = OracleFeaturePropagation(150 + extra, [128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_parts, 1)
        self.extra_conv_emb = nn.Conv1d(128, 128, 1)

    def forward(self, xyz, cls_label, chamfer_points=None, include_convex_loss=False, quantile=0.01,
                msc_iterations=5, max_num_clusters=25, fps_start=None, fit_inputs=None, **_unused):
        B, _, N = xyz.shape
        l0_points = xyz
        l0_xyz = xyz[:, :3, :] if self.normal_channel else xyz
        s1, s2 = (fps_start if fps_start is not None else (None, None))
        l1_xyz, l1_points = self.sa1(l0_xyz, l0_points, s1)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, s2)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        onehot = cls_label.view(B, 16, 1).repeat(1, 1, N)
        l0_points = self.fp1(l0_xyz, l1_xyz, torch.cat([onehot, l0_xyz, l0_points], 1), l1_points)
        x = F.conv1d(l0_points, self.conv1.weight, self.conv1.bias)
        feat = F.relu(F.batch_norm(x, self.bn1.running_mean, self.bn1.running_var, self.bn1.weight,
                                   self.bn1.bias, self.bn1.training, self.bn1.momentum, self.bn1.eps))
        total = torch.zeros(1)
        chamfer = torch.zeros(1)
        extra = ()
        if include_convex_loss:
            if self.beta > 0.001:
                self.beta *= 0.99
            emb = F.conv1d(feat, self.extra_conv_emb.weight, self.extra_conv_emb.bias)
            total, chamfer, params, labels = convex_loss(
                xyz, chamfer_points, emb, quantile=quantile, iterations=msc_iterations,
                max_num_clusters=max_num_clusters, **(fit_inputs or {}))
            extra = (labels, params, emb)
        logits = F.conv1d(self.drop1(feat), self.conv2.weight, self.conv2.bias)
        seg = F.log_softmax(logits, dim=1).transpose(1, 2)
        return (seg, (l1_points, l2_points, l3_points), feat, total, chamfer) + extra


class OracleSSGPartSeg(nn.Module):
    """models/pointnet2_part_seg_ssg.py:7-49 (config 1 plumbing case): returns (log-probs, l3)."""

    def __init__(self, num_classes, normal_channel=False):
        super().__init__()
        extra = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.sa1 = OracleSetAbstraction(512, 0.2, 32, 6 + extra, [64, 64, 128], False)
        self.sa2 = OracleSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = OracleSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True)
        self.fp3 = OracleFeaturePropagation(1280, [256, 256])
        self.fp2 = OracleFeaturePropagation(384, [256, 128])
        self.fp1 = OracleFeaturePropagation(128 + 16 + 6 + extra, [128, 128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def forward(self, xyz, cls_label, fps_start=None):
        B, _, N = xyz.shape
        l0_points = xyz
        l0_xyz = xyz[:, :3, :] if self.normal_channel else xyz
        s1, s2 = (fps_start if fps_start is not None else (None, None))
        l1_xyz, l1_points = self.sa1(l0_xyz, l0_points, s1)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, s2)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        onehot = cls_label.view(B, 16, 1).repeat(1, 1, N)
        l0_points = self.fp1(l0_xyz, l1_xyz, torch.cat([onehot, l0_xyz, l0_points], 1), l1_points)
        x = F.conv1d(l0_points, self.conv1.weight, self.conv1.bias)
        feat = F.relu(F.batch_norm(x, self.bn1.running_mean, self.bn1.running_var, self.bn1.weight,
                                   self.bn1.bias, self.bn1.training, self.bn1.momentum, self.bn1.eps))
        logits = F.conv1d(self.drop1(feat), self.conv2.weight, self.conv2.bias)
        return F.log_softmax(logits, dim=1).transpose(1, 2), l3_points


def seg_loss(pred, target):
    """models/pointnet2_part_seg_msg.py:137-144: cross_entropy applied to log-probabilities."""
    return F.cross_entropy(pred, target)


def convex_loss(*a, **k):  # filled in by the fitting half of the oracle (below)
    raise NotImplementedError
