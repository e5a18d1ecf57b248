"""PointNet++ SSG part-segmentation network (BASELINE.json configs[0], the plumbing case) on the
MI355X backend: call surface of the reference's models/pointnet2_part_seg_ssg.py:7-58
(`get_model(num_classes, normal_channel)`, `forward(xyz, cls_label) -> (log-probs [B,N,C], l3_points)`,
`get_loss` = F.nll_loss)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn_ops import LinearFn, SharedMLPFn
from .pointnet_util import PointNetFeaturePropagation, PointNetSetAbstraction, _mlp_cfg, _mlp_tensors


class get_model(nn.Module):
    def __init__(self, num_classes, normal_channel=False):
        super().__init__()
        additional_channel = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=6 + additional_channel,
                                          mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3,
                                          mlp=[128, 128, 256], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3,
                                          mlp=[256, 512, 1024], group_all=True)
        self.fp3 = PointNetFeaturePropagation(in_channel=1280, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=384, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=128 + 16 + 6 + additional_channel, mlp=[128, 128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def forward(self, xyz, cls_label, fps_start=None):
        B, C, N = xyz.shape
        pts = xyz.permute(0, 2, 1).contiguous()
        l0_xyz = pts[:, :, :3].contiguous() if self.normal_channel else pts
        s1, s2 = fps_start if fps_start is not None else (None, None)
        l1_xyz, l1_points = self.sa1.forward_cl(l0_xyz, pts, s1)
        l2_xyz, l2_points = self.sa2.forward_cl(l1_xyz, l1_points, s2)
        l3_xyz, l3_points = self.sa3.forward_cl(l2_xyz, l2_points)
        l2_up = self.fp3.forward_cl(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_up = self.fp2.forward_cl(l1_xyz, l2_xyz, l1_points, l2_up)
        onehot = cls_label.reshape(B, 1, 16).expand(B, N, 16)
        l0_up = self.fp1.forward_cl(l0_xyz, l1_xyz, torch.cat([onehot, l0_xyz, pts], dim=-1), l1_up)
        feat = SharedMLPFn.apply(l0_up.reshape(B * N, -1), _mlp_cfg([self.bn1], 0, self.training),
                                 *_mlp_tensors([self.conv1], [self.bn1], self.conv1.weight.reshape(128, 128)))
        logits = LinearFn.apply(self.drop1(feat), self.conv2.weight.reshape(self.conv2.weight.shape[0], 128),
                                self.conv2.bias)
        return F.log_softmax(logits, dim=1).reshape(B, N, -1), l3_points.permute(0, 2, 1)


class get_loss(nn.Module):
    def forward(self, pred, target, trans_feat=None):
        return F.nll_loss(pred, target)
