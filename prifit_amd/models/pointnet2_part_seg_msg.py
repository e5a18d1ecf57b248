"""PointNet++ MSG part-segmentation network on the MI355X backend.

Call surface of the reference's models/pointnet2_part_seg_msg.py (get_model :11-134, get_loss
:137-144, get_selfsup_loss :147-171) and of models/pretrain_pointnet2_part_seg_msg.py: same
constructor arguments, forward keyword arguments, sub-module / parameter names (checkpoint
compatible) and output tensors.

Return contract (fixes the upstream arity bug): always the trainer's 5-tuple
    (seg_logprob [B,N,num_parts], (l1_points, l2_points, l3_points), feat [B,128,N],
     total_loss [1,1] or [1], chamfer_loss [1,1] or [1])
(train_partseg_shapenet.py:387,444); when include_convex_loss=True the tuple is extended by
(labels, ellipse_params_batch, feat_embed) as in upstream :134.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn_ops import LinearFn, SharedMLPFn, cross_entropy
from .. import arena as zero_pool
from .pointnet_util import (pack_plan, PointNetFeaturePropagation, PointNetSetAbstraction, PointNetSetAbstractionMsg,
                            _mlp_cfg, _mlp_tensors, batched_bn_counters)


class get_model(nn.Module):
    def __init__(self, num_parts, normal_channel=False, l2_norm=False, reconstruct=False, extra_layers=False,
                 num_charts=25, num_points=128):
        super().__init__()
        if reconstruct or extra_layers:
            raise NotImplementedError("reconstruct / extra_layers are outside the accelerated hot path")
        additional_channel = 3 if normal_channel else 0
        self.normal_channel = normal_channel
        self.l2_norm = l2_norm
        self.num_charts, self.num_points = num_charts, num_points
        self.beta = 1
        self.extra_layers = extra_layers
        self.reconstruct = reconstruct
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], 3 + additional_channel,
                                             [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 128 + 128 + 64,
                                             [[128, 128, 256], [128, 196, 256]])
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=512 + 3,
                                          mlp=[256, 512, 1024], group_all=True)
        self.fp3 = PointNetFeaturePropagation(in_channel=1536, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=576, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=150 + additional_channel, mlp=[128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_parts, 1)
        self.extra_conv_emb = nn.Conv1d(128, 128, 1)

    def embed_features(self, xyz, cls_label, fps_start=None):
        """Backbone up to `feat` (upstream :64-88), channels-last internally.
        Returns (l1 [B,512,320], l2 [B,128,256], l3 [B,1,1024], feat [B*N,128]) channels-last."""
        with batched_bn_counters():
            return self._embed_features(xyz, cls_label, fps_start)

    def sample_ahead(self, xyz, fps_start=None):
        """The two farthest-point-sampling levels of a batch `xyz` [B,C,N] on a side stream (ops.sample_ahead): call it
        for the NEXT batch before running the current one and pass the result as that step's `fps_start`."""
        from .. import ops
        pts = xyz.permute(0, 2, 1)
        l0_xyz = (pts[:, :, :3] if self.normal_channel else pts).contiguous()
        return tuple(ops.sample_ahead(l0_xyz, (self.sa1.npoint, self.sa2.npoint), fps_start))

    def _embed_features(self, xyz, cls_label, fps_start=None):
        B, C, N = xyz.shape
        pts = xyz.permute(0, 2, 1).contiguous()           # l0_points (= xyz, also without normals: :69-75)
        l0_xyz = pts[:, :, :3].contiguous() if self.normal_channel else pts
        s1, s2 = fps_start if fps_start is not None else (None, None)
        with pack_plan(self):    # the column-packed first-layer weights of every module: one launch (and one in the backward)
            return self._embed_layers(B, N, pts, l0_xyz, cls_label, s1, s2)

    def _embed_layers(self, B, N, pts, l0_xyz, cls_label, s1, s2):
        l1_xyz, l1_points = self.sa1.forward_cl(l0_xyz, pts, s1)
        l2_xyz, l2_points = self.sa2.forward_cl(l1_xyz, l1_points, s2)
        l3_xyz, l3_points = self.sa3.forward_cl(l2_xyz, l2_points)
        l2_up = self.fp3.forward_cl(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_up = self.fp2.forward_cl(l1_xyz, l2_xyz, l1_points, l2_up)
        onehot = cls_label.reshape(B, 1, 16).expand(B, N, 16)
        skip = torch.cat([onehot, l0_xyz, pts], dim=-1)  # :86 [onehot16, xyz, points]
        l0_up = self.fp1.forward_cl(l0_xyz, l1_xyz, skip, l1_up)
        w1 = self.conv1.weight.reshape(128, 128)
        feat = SharedMLPFn.apply(l0_up.reshape(B * N, -1), _mlp_cfg([self.bn1], 0, self.training),
                                 *_mlp_tensors([self.conv1], [self.bn1], w1))
        return l1_up, l2_up, l3_points, feat

    def forward(self, xyz, cls_label, chamfer_points=0, include_convex_loss=False, if_cuboid=False,
                include_intersect_loss=False, include_entropy_loss=False, include_pruning=False, quantile=0.01,
                msc_iterations=5, max_num_clusters=25, visualize=False, seed=0, batch_id=0, class_list=[],
                epoch=-1, alpha=1, beta=1, evaluation=False, embed=False, fps_start=None, fit_inputs=None):
        B, C, N = xyz.shape
        if xyz.is_cuda:
            zero_pool.begin_step(xyz.device)   # one zero-fill per step for all zero-initialised fp32 buffers
        l1, l2, l3, feat = self.embed_features(xyz, cls_label, fps_start)
        if getattr(self, "after_backbone", None) is not None:
            # data-path hook (like a DataLoader worker): e.g. `net.after_backbone = lambda: net.sample_ahead(next_xyz)` starts
            # the NEXT batch's farthest-point sampling here, so that its 640 serial rounds on 24 CUs run beside the
            # matrix-bound mean-shift kernels of this step instead of at the head of the next one
            self.after_backbone()
        total_loss = chamfer_loss = None                    # zeros(1) each unless the convex loss sets them (:96-97)
        extra = ()
        feat_embed = None
        if embed and not include_convex_loss:
            feat_embed = LinearFn.apply(feat, self.extra_conv_emb.weight.reshape(128, 128), self.extra_conv_emb.bias)
        if include_convex_loss:
            from ..convex_loss import convex_loss

            if self.beta > 0.001:
                self.beta *= 0.99
            else:
                include_entropy_loss = False
            emb = LinearFn.apply(feat, self.extra_conv_emb.weight.reshape(128, 128), self.extra_conv_emb.bias)
            if self.l2_norm:
                emb = F.normalize(emb, p=2, dim=1)
            feat_embed = emb.reshape(B, N, 128).permute(0, 2, 1)
            total_loss, chamfer_loss, params, labels = convex_loss(
                xyz, chamfer_points, feat_embed, if_cuboid=if_cuboid, quantile=quantile,
                include_pruning=include_pruning, include_intersect_loss=include_intersect_loss,
                include_entropy_loss=include_entropy_loss, iterations=msc_iterations,
                max_num_clusters=max_num_clusters, visualize=visualize, seed=seed, batch_id=batch_id, epoch=epoch,
                class_list=class_list, alpha=alpha, beta=self.beta, evaluation=evaluation,
                **(fit_inputs or {}))
            extra = (labels, params, feat_embed)
        elif embed:
            extra = (None, None, feat_embed.reshape(B, N, 128).permute(0, 2, 1))
        if total_loss is None:
            total_loss, chamfer_loss = torch.zeros(1, device=xyz.device), torch.zeros(1, device=xyz.device)
        x = self.drop1(feat)
        logits = LinearFn.apply(x, self.conv2.weight.reshape(self.conv2.weight.shape[0], 128), self.conv2.bias)
        seg = F.log_softmax(logits, dim=1).reshape(B, N, -1)
        feat_cf = feat.reshape(B, N, 128).permute(0, 2, 1)
        outs = (seg, (l1.permute(0, 2, 1), l2.permute(0, 2, 1), l3.permute(0, 2, 1)), feat_cf, total_loss,
                chamfer_loss)
        return outs + extra


class get_loss(nn.Module):
    """upstream :137-144: F.cross_entropy on log-probabilities (softmax applied twice: kept)."""

    def forward(self, pred, target, trans_feat=None):
        return cross_entropy(pred, target)       # (F.cross_entropy; on the GPU one pass each way, nn_ops.CrossEntropyFn)


class get_selfsup_loss(nn.Module):
    """upstream :147-171 (contrastive pair loss; not on the benchmarked path, plain torch ops)."""

    def __init__(self, margin=0.5):
        super().__init__()
        self.margin = margin

    def forward(self, feat, target):
        feat = F.normalize(feat, p=2, dim=1)
        pair_sim = torch.bmm(feat.transpose(1, 2), feat)
        onehot = F.one_hot(target).float()
        pair_target = torch.bmm(onehot, onehot.transpose(1, 2))
        loss = pair_target * (1.0 - pair_sim) + (1.0 - pair_target) * F.relu(pair_sim - self.margin)
        diag_mask = 1 - torch.eye(loss.shape[-1], device=loss.device)
        with torch.no_grad():
            pos_fraction = (pair_target == 1).float().mean()
            sample_neg = torch.rand_like(pair_target) > 1 - pos_fraction
            sample_mask = (pair_target == 1) | sample_neg
        loss = diag_mask.unsqueeze(0) * sample_mask.float() * loss
        return 0.5 * loss.mean()
