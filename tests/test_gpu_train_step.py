"""GPU: the training iteration (train_partseg_shapenet.py:372-399 / :436-451) on the HIP backend vs the same
iteration of the oracle model with torch's Adam: parameters after one supervised step, schedule, checkpoint."""
import os
import tempfile

import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import synth
from tests_helpers import fit_inputs

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_supervised_step_matches_oracle_adam(hiplib):
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.train_step import Trainer
    B, N = 2, 1024
    torch.manual_seed(3)
    ref = orc.OracleMSGPartSeg(50)
    synth.xavier_like_trainer(ref)
    net = M.get_model(50)
    net.load_state_dict(ref.state_dict())
    net.cuda()
    for n in (ref, net):
        n.train()
        n.drop1.eval()
    pts = _t(synth.cloud("surface", B, N, 5))
    target = _t(synth.labels(B, N, 50, 5))
    s = (_t(synth.fps_start(B, N, 5)), _t(synth.fps_start(B, 512, 6)))
    # oracle: one Adam step exactly as upstream :252-259, :396-399
    opt = torch.optim.Adam(ref.parameters(), lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    opt.zero_grad()
    seg = ref(pts.transpose(2, 1).contiguous(), torch.zeros(B, 1, 16), fps_start=s)[0]
    loss_ref = orc.seg_loss(seg.reshape(-1, 50), target.view(-1))
    loss_ref.backward()
    opt.step()
    tr = Trainer(net, learning_rate=0.001, decay_rate=1e-4)
    net.drop1.eval()
    real_train = net.train
    net.train = lambda mode=True: (real_train(mode), net.drop1.eval(), net)[2]  # keep dropout off for the comparison
    loss, acc = tr.supervised_step(pts.cuda(), target.cuda(), augment=False, fps_start=(s[0].cuda(), s[1].cuda()))
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item()) and 0 <= acc.item() <= 1
    # Adam's first step moves every weight by ~lr*sign(g): compare the parameter UPDATE directions on the large gradients
    for (k, p_ref), (_, p) in zip(ref.named_parameters(), net.named_parameters()):
        torch.testing.assert_close(p.detach().cpu(), p_ref.detach(), rtol=0, atol=2.5e-3, msg=lambda m: k + ": " + m)  # |update| <= lr*(1+wd)
        if p_ref.grad is None:  # extra_conv_emb is not on the supervised path
            continue
        big = p_ref.grad.abs() > 0.1 * p_ref.grad.abs().max()
        if k in ("conv2.weight", "conv1.weight", "fp1.mlp_convs.1.weight") and big.any():
            d = (p.detach().cpu() - p_ref.detach()).abs()[big]
            assert d.max() < 2e-4, (k, d.max())
    lr, mom = tr.set_epoch(45)
    assert abs(lr - 0.001 * 0.5 ** 2) < 1e-12 and abs(mom - 0.1 * 0.25) < 1e-12 and abs(net.bn1.momentum - 0.025) < 1e-12
    lr, mom = tr.set_epoch(400)
    assert lr == 1e-5 and mom == 0.01


def test_selfsup_step_and_checkpoint(hiplib):
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.train_step import Trainer
    _, cham, _ = fit_inputs(2, 2048, 128, 4)
    torch.manual_seed(4)
    net = M.get_model(50).cuda()
    tr = Trainer(net, lmbda=1.0)
    w0 = net.extra_conv_emb.weight.detach().clone()
    ss = tr.selfsup_step(cham.cuda(), quantile=0.05, msc_iterations=10, max_num_clusters=25)
    assert torch.isfinite(ss) and not torch.equal(net.extra_conv_emb.weight.detach(), w0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "model_001.pth")
        tr.save(path)
        ck = torch.load(path, map_location="cpu")
        assert set(ck) == {"epoch", "train_acc", "model_state_dict", "optimizer_state_dict"}  # upstream :468-473
        net2 = M.get_model(50).cuda()
        Trainer(net2).load(path)
        for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
            assert torch.equal(a.cpu(), b.cpu())
