"""Same network as models.pointnet2_part_seg_msg; upstream keeps a second copy whose forward returns
the trainer's 5-tuple (models/pretrain_pointnet2_part_seg_msg.py:39-88).  Here both names resolve to
one implementation that always honours that contract."""
from .pointnet2_part_seg_msg import get_loss, get_model, get_selfsup_loss  # noqa: F401
