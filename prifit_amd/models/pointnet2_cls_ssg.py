"""PointNet++ SSG classification network on the MI355X backend: call surface of the reference's
models/pointnet2_cls_ssg.py:7-40."""
import torch.nn as nn
import torch.nn.functional as F

from ..nn_ops import LinearFn
from .pointnet2_cls_msg import _fc_bn_relu, get_loss  # noqa: F401
from .pointnet_util import PointNetSetAbstraction


class get_model(nn.Module):
    def __init__(self, num_class, normal_channel=True):
        super().__init__()
        in_channel = 6 if normal_channel else 3
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=in_channel, mlp=[64, 64, 128],
                                          group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256],
                                          group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3,
                                          mlp=[256, 512, 1024], group_all=True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
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
