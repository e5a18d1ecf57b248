"""PointNet++ semantic-segmentation network on the MI355X backend: call surface of the reference's
models/pointnet2_sem_seg.py:6-48 (`get_model(num_classes, with_rgb)`, `forward(xyz [B,C,N]) -> (log-probs
[B,N,num_classes], l4_points [B,512,16])`, `get_loss(pred, target, trans_feat, weight)` = weighted nll_loss)."""
import torch.nn as nn
import torch.nn.functional as F

from ..nn_ops import LinearFn, SharedMLPFn
from .pointnet_util import PointNetFeaturePropagation, PointNetSetAbstraction, _mlp_cfg, _mlp_tensors


class get_model(nn.Module):
    def __init__(self, num_classes, with_rgb=True):
        super().__init__()
        self.with_rgb = with_rgb
        additional_channel = 3 if with_rgb else 0
        self.sa1 = PointNetSetAbstraction(1024, 0.1, 32, 6 + additional_channel, [32, 32, 64], False)
        self.sa2 = PointNetSetAbstraction(256, 0.2, 32, 64 + 3, [64, 64, 128], False)
        self.sa3 = PointNetSetAbstraction(64, 0.4, 32, 128 + 3, [128, 128, 256], False)
        self.sa4 = PointNetSetAbstraction(16, 0.8, 32, 256 + 3, [256, 256, 512], False)
        self.fp4 = PointNetFeaturePropagation(768, [256, 256])
        self.fp3 = PointNetFeaturePropagation(384, [256, 256])
        self.fp2 = PointNetFeaturePropagation(320, [256, 128])
        self.fp1 = PointNetFeaturePropagation(128, [128, 128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def forward(self, xyz, fps_start=None):
        B, C, N = xyz.shape
        pts = xyz.permute(0, 2, 1).contiguous()
        l0_xyz = pts[:, :, :3].contiguous() if self.with_rgb else pts
        s = fps_start if fps_start is not None else (None, None, None, None)
        l1_xyz, l1_points = self.sa1.forward_cl(l0_xyz, pts, s[0])
        l2_xyz, l2_points = self.sa2.forward_cl(l1_xyz, l1_points, s[1])
        l3_xyz, l3_points = self.sa3.forward_cl(l2_xyz, l2_points, s[2])
        l4_xyz, l4_points = self.sa4.forward_cl(l3_xyz, l3_points, s[3])
        l3_up = self.fp4.forward_cl(l3_xyz, l4_xyz, l3_points, l4_points)
        l2_up = self.fp3.forward_cl(l2_xyz, l3_xyz, l2_points, l3_up)
        l1_up = self.fp2.forward_cl(l1_xyz, l2_xyz, l1_points, l2_up)
        l0_up = self.fp1.forward_cl(l0_xyz, l1_xyz, None, l1_up)
        feat = SharedMLPFn.apply(l0_up.reshape(B * N, -1), _mlp_cfg([self.bn1], 0, self.training),
                                 *_mlp_tensors([self.conv1], [self.bn1], self.conv1.weight.reshape(128, 128)))
        logits = LinearFn.apply(self.drop1(feat), self.conv2.weight.reshape(self.conv2.weight.shape[0], 128),
                                self.conv2.bias)
        return F.log_softmax(logits, dim=1).reshape(B, N, -1), l4_points.permute(0, 2, 1)


class get_loss(nn.Module):
    def forward(self, pred, target, trans_feat=None, weight=None):
        return F.nll_loss(pred, target, weight=weight)
