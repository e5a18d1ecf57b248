"""PointNet++ MSG classification network on the MI355X backend: call surface of the reference's
models/pointnet2_cls_msg.py:7-41 (`get_model(num_class, normal_channel)`, `forward(xyz) -> (log-probs [B,C],
l3_points [B,1024,1])`, `get_loss` = F.nll_loss)."""
import torch.nn as nn
import torch.nn.functional as F

from ..nn_ops import LinearFn, SharedMLPFn
from .pointnet_util import PointNetSetAbstraction, PointNetSetAbstractionMsg, _mlp_cfg, _mlp_tensors


def _fc_bn_relu(x, fc, bn, training):
    """relu(bn(fc(x))) on [B, C] rows: one GEMM with the BatchNorm statistics in its epilogue."""
    return SharedMLPFn.apply(x, _mlp_cfg([bn], 0, training), *_mlp_tensors([fc], [bn], fc.weight))


class get_model(nn.Module):
    def __init__(self, num_class, normal_channel=True):
        super().__init__()
        in_channel = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], in_channel,
                                             [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320,
                                             [[64, 64, 128], [128, 128, 256], [128, 128, 256]])
        self.sa3 = PointNetSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.5)
        self.fc3 = nn.Linear(256, num_class)

    def forward(self, xyz, fps_start=None):
        B = xyz.shape[0]
        pts = xyz.permute(0, 2, 1).contiguous()
        norm = pts[:, :, 3:].contiguous() if self.normal_channel else None
        l0_xyz = pts[:, :, :3].contiguous()
        s1, s2 = fps_start if fps_start is not None else (None, None)
        l1_xyz, l1_points = self.sa1.forward_cl(l0_xyz, norm, s1)
        l2_xyz, l2_points = self.sa2.forward_cl(l1_xyz, l1_points, s2)
        _, l3_points = self.sa3.forward_cl(l2_xyz, l2_points)
        x = l3_points.reshape(B, 1024)
        x = self.drop1(_fc_bn_relu(x, self.fc1, self.bn1, self.training))
        x = self.drop2(_fc_bn_relu(x, self.fc2, self.bn2, self.training))
        x = LinearFn.apply(x, self.fc3.weight, self.fc3.bias)
        return F.log_softmax(x, -1), l3_points.permute(0, 2, 1)


class get_loss(nn.Module):
    def forward(self, pred, target, trans_feat=None):
        return F.nll_loss(pred, target)
