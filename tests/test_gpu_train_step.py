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


@pytest.mark.parametrize("split", ["0", "bf16x6", "fp16x3"])
def test_selfsup_step_matches_reference_golden(hiplib, golden, split, monkeypatch):
    """(split != "0": the labelled 16-bit-products experiment of the mean-shift forward, fit_ops.MS_SPLIT, at the same bars.)
    SURVEY 8a row a29 step (2) on the HIP backend, through Trainer.selfsup_step itself (zero_grad, train(), forward
    with the convex loss, mean(loss) * lambda, backward, Adam): total / chamfer loss, K, labels, beta, every
    parameter-gradient norm, the embedding head's gradient and its values after the Adam step against what the
    reference's own network file produced (tests/golden/step_selfsup.npz, train_partseg_shapenet.py:436-451)."""
    import step_selfsup_common as C
    from prifit_amd import fit_ops
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.train_step import Trainer
    monkeypatch.setattr(fit_ops, "MS_SPLIT", split)
    launches0 = fit_ops.split_launches
    g = golden("step_selfsup")
    d = C.inputs(g)
    ref = C.seeded_state(g, orc.OracleMSGPartSeg)          # seeded parameters (construction order of the reference)
    net = M.get_model(50)
    net.load_state_dict(ref.state_dict())
    net.cuda()
    tr = Trainer(net, learning_rate=0.001, decay_rate=1e-4, lmbda=1.0)
    real_train = net.train
    net.train = lambda mode=True: (real_train(mode), net.drop1.eval(), net)[2]
    before = {k: p.detach().cpu().clone() for k, p in net.named_parameters()}
    captured = {}
    real_forward = net.forward

    def forward(*a, **k):                                    # keep the model's outputs and gradients of this step
        out = real_forward(*a, **k)
        captured["out"] = out
        return out

    net.forward = forward
    real_step = tr.optimizer.step

    def step(*a, **k):
        captured["grads"] = {kk: (None if p.grad is None else p.grad.detach().cpu().clone()) for kk, p in net.named_parameters()}
        return real_step(*a, **k)

    tr.optimizer.step = step
    subset = torch.from_numpy(np.random.default_rng(int(g["seed"]) + 1).choice(5000, C.N, replace=False)).cuda()
    loss = tr.selfsup_step(d["cham"].transpose(1, 2).contiguous().cuda(), npoint=C.N, quantile=C.Q, msc_iterations=C.ITERS,
                           max_num_clusters=25, augment=False, subset=subset, fps_start=(d["s1"].cuda(), d["s2"].cuda()),
                           fit_inputs=dict(rand_table=d["R"].cuda(), canonical=True, center_ids=d["center_ids"]))
    after = {k: p.detach().cpu().clone() for k, p in net.named_parameters()}
    assert fit_ops.split_launches - launches0 == (C.ITERS if split != "0" else 0)
    assert abs(loss.item() - float(np.asarray(g["total_loss"]).reshape(-1)[0])) < 1e-4 * abs(loss.item())
    worst = C.check(g, captured["out"], captured["grads"], before, after, net.beta, loss_tol=1e-4, grad_tol=2e-2)
    print("self-supervised step vs reference: worst relative gradient deviation %.2e" % worst)


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


def test_speculative_retry_restores_python_state(hiplib):
    """A step whose clustering verdict asks for the quantile-doubling retry is discarded and re-run: BatchNorm buffers,
    the entropy-weight schedule `beta *= 0.99` (models/pointnet2_part_seg_msg.py:96-99: once per ACCEPTED step) and the
    device RNG stream are put back first."""
    from prifit_amd import fit_ops
    from prifit_amd.train_step import SpeculativeRunner

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = torch.nn.BatchNorm1d(4)
            self.beta = 1.0

    m = Toy().cuda()
    runner = SpeculativeRunner(m)
    calls = []

    def fn():
        m.beta *= 0.99
        m.bn(torch.ones(3, 4, device="cuda") * (len(calls) + 1))          # moves the running statistics
        draw = torch.rand(4, device="cuda")
        if fit_ops._spec is not None:                                     # first (speculative) attempt: verdict "retry"
            flag = torch.ones(1, dtype=torch.int32).pin_memory()
            ev = torch.cuda.Event()
            ev.record()
            fit_ops._spec.checks.append((ev, flag))
        calls.append((m.bn.running_mean.clone(), draw))
        return draw

    rm0 = m.bn.running_mean.clone()
    runner.run(fn, lambda: None)
    assert len(calls) == 2 and runner.fallbacks == 1
    assert abs(m.beta - 0.99) < 1e-12                                     # advanced once, not twice
    assert torch.equal(calls[0][1], calls[1][1])                          # same random draw in the retried step
    expect = rm0 * 0.9 + 0.1 * 2.0                                        # one momentum update, from the second call's input
    torch.testing.assert_close(m.bn.running_mean, expect)


def test_graphed_backbone_matches_eager(hiplib):
    """train_step.graph_backbone: the backbone's forward and backward replayed as two HIP graphs give the eager step's loss
    and parameter gradients (supervised step, models/pointnet2_part_seg_msg.py:64-88 + autograd), BatchNorm buffers are
    left as the eager step leaves them, and a second replay with other input VALUES (same shapes) follows the eager path."""
    import numpy as np
    from prifit_amd import synth
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.train_step import graph_backbone
    dev = torch.device("cuda", 0)
    B, N = 2, 1024

    def batch(seed):
        xyz = torch.from_numpy(synth.cloud("surface", B, N, seed)).transpose(1, 2).contiguous().to(dev)
        tgt = torch.from_numpy(synth.labels(B, N, 50, seed)).to(dev)
        return xyz, tgt

    starts = (torch.from_numpy(synth.fps_start(B, N, 3)).to(dev), torch.from_numpy(synth.fps_start(B, 512, 4)).to(dev))
    cls = torch.zeros(B, 1, 16, device=dev)

    def make():
        torch.manual_seed(0)
        net = M.get_model(50)
        synth.xavier_like_trainer(net)
        net = net.to(dev).train()
        net.drop1.eval()
        return net

    def run(net, xyz, tgt):
        for p in net.parameters():
            p.grad = None
        seg = net(xyz, cls, fps_start=starts)[0]
        loss = torch.nn.functional.cross_entropy(seg.reshape(-1, 50), tgt.view(-1))
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    eager, graphed = make(), make()
    graph_backbone(graphed, *batch(5)[:1], cls, starts)
    for n, b in eager.named_buffers():                      # the capture's warm-up iterations left no trace
        assert torch.equal(b, dict(graphed.named_buffers())[n]), n
    for seed in (5, 6):
        xyz, tgt = batch(seed)
        le, ge = run(eager, xyz, tgt)
        lg, gg = run(graphed, xyz, tgt)
        assert abs(le - lg) <= 1e-6 * abs(le)
        assert set(ge) == set(gg)
        for n in ge:
            if ge[n].norm() > 1e-6:                          # (bias gradients in front of a BatchNorm are rounding noise)
                assert (ge[n] - gg[n]).norm() <= 2e-3 * ge[n].norm(), (n, (ge[n] - gg[n]).norm().item(), ge[n].norm().item())
    for n, b in eager.named_buffers():
        torch.testing.assert_close(b, dict(graphed.named_buffers())[n], rtol=1e-5, atol=1e-6)
    # an evaluation forward must NOT replay the training-mode graph (batch statistics, running-stat updates): the installed
    # embed_features falls back to the eager backbone in eval mode / under no_grad (advisor, round 3)
    xyz, _ = batch(7)
    eager.eval()
    graphed.eval()
    snap = {n: b.clone() for n, b in graphed.named_buffers()}
    with torch.no_grad():
        se = eager(xyz, cls, fps_start=starts)[0]
        sg = graphed(xyz, cls, fps_start=starts)[0]
    torch.testing.assert_close(sg, se, rtol=1e-5, atol=1e-5)
    for n, b in graphed.named_buffers():
        assert torch.equal(b, snap[n]), n                    # running statistics untouched


def test_prefetched_selfsup_step_equals_inline_step(hiplib):
    """Trainer.prefetch_selfsup prepares the NEXT batch (augmentation off here, explicit subset) and starts its
    farthest-point sampling on a side stream behind the backbone forward of the step that runs in between; the following
    selfsup_step() consumes it.  Same loss and the same parameters afterwards as the in-line step on an identical model:
    the samples are the ones the in-line launch computes (the headline's FPS placement is something the
    trainer can do)."""
    from tests_helpers import fit_inputs
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.ops import SampledAhead
    from prifit_amd.train_step import Trainer
    B, N, Mpts = 2, 1024, 2500
    _, cham, _ = fit_inputs(B, N, 128, 8, M=Mpts)
    cham = cham.cuda()
    _, cham0, _ = fit_inputs(B, N, 128, 9, M=Mpts)
    cham0 = cham0.cuda()
    subset = torch.from_numpy(np.random.default_rng(3).choice(Mpts, N, replace=False)).cuda()
    starts = (torch.from_numpy(synth.fps_start(B, N, 5)).cuda(), torch.from_numpy(synth.fps_start(B, 512, 6)).cuda())
    kw = dict(npoint=N, quantile=0.05, msc_iterations=5, max_num_clusters=25)
    res = []
    for prefetch in (True, False):
        torch.manual_seed(11)
        net = M.get_model(50)
        synth.xavier_like_trainer(net)
        net.cuda()
        tr = Trainer(net)
        torch.manual_seed(12)                      # dropout / covariance-noise streams identical in both arms
        if prefetch:
            tr.prefetch_selfsup(cham, npoint=N, augment=False, subset=subset, fps_start=starts)
            assert net.after_backbone is not None
        l0 = tr.selfsup_step(cham0, augment=False, subset=subset, fps_start=starts, **kw)      # the step in between
        if prefetch:
            assert net.after_backbone is None and isinstance(tr._next["ahead"][0], SampledAhead)   # launched behind its backbone
            l1 = tr.selfsup_step(**kw)
            assert tr._next is None
        else:
            l1 = tr.selfsup_step(cham, augment=False, subset=subset, fps_start=starts, **kw)
        tr.finish()
        res.append((l0.item(), l1.item(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()))
    (a0, a1, pa), (b0, b1, pb) = res
    assert abs(a0 - b0) <= 1e-5 * abs(b0) and abs(a1 - b1) <= 1e-4 * abs(b1), (a0, b0, a1, b1)
    torch.testing.assert_close(pa, pb, rtol=1e-3, atol=1e-5)
    with pytest.raises(RuntimeError):
        tr.selfsup_step(**kw)                      # nothing prefetched


def test_alternating_trainer_steps_train(hiplib):
    """The reference's loop (train_partseg_shapenet.py:346-451): supervised step, then self-supervised step, several times on
    one synthetic batch through Trainer -- prefetching the self-supervised batch each time.  Everything stays finite, the
    supervised loss goes down, every parameter has moved, and the optimizer has stepped each parameter 2 x iterations times
    (zero-filled gradients for the parameters a step does not reach: the reference's zero_grad semantics)."""
    from tests_helpers import fit_inputs
    from prifit_amd.models import pointnet2_part_seg_msg as M
    from prifit_amd.train_step import Trainer
    B, N, Mpts, iters = 4, 1024, 2500, 5
    _, cham, _ = fit_inputs(B, N, 128, 21, M=Mpts)
    cham = cham.cuda()
    sup_pts = cham[:, :N].contiguous()
    target = torch.from_numpy(synth.part_labels(sup_pts.cpu().numpy(), 8, 3)).long().cuda()     # spatial parts: learnable
    torch.manual_seed(5)
    np.random.seed(5)
    net = M.get_model(50)
    synth.xavier_like_trainer(net)
    net.cuda()
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    tr = Trainer(net, learning_rate=0.002)
    sup, ss = [], []
    for it in range(iters):
        tr.prefetch_selfsup(cham, npoint=N)
        loss, acc = tr.supervised_step(sup_pts, target)
        sup.append(loss.item())
        ss.append(tr.selfsup_step(quantile=0.05, msc_iterations=5, max_num_clusters=25).item())
    tr.finish()
    assert all(np.isfinite(sup)) and all(np.isfinite(ss)), (sup, ss)
    assert sup[-1] < 0.9 * sup[0], sup
    for k, p in net.named_parameters():
        assert torch.isfinite(p).all(), k
        # (a conv bias in front of a batch-statistics BatchNorm has an exactly zero gradient and starts at zero: it stays there)
        if k.endswith("weight"):
            assert not torch.equal(p.detach(), before[k]), k
    steps = {int(st["step"]) for st in tr.optimizer.state.values()}
    # conv2 first gets a gradient in step 1 (supervised), extra_conv_emb in step 2 (self-supervised): afterwards every
    # parameter steps every time
    assert steps <= {2 * iters, 2 * iters - 1}, steps


def test_flat_adam_matches_torch_adam(hiplib):
    """prifit_adam_flat (one launch over the flat parameter buffer, prifit_amd/optim.py) against torch.optim.Adam (the
    single-tensor CPU path, in fp32 and in fp64) on the same gradients: ragged tensor sizes (a length that is not a multiple of
    four, a scalar), a parameter that gets its first gradient at step 3 and one that never does (skipped: no decay, no step
    count), a learning-rate change in between, a step discarded through the device `skip` flag, and the checkpoint format in both
    directions (train_partseg_shapenet.py:252-259, :398, :467-475)."""
    from prifit_amd.optim import FlatAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    shapes = [(64, 9), (50,), (1,), (128, 64, 1, 1), (7, 3), (33,), (256, 131)]
    mk = lambda dt, d: [torch.nn.Parameter(torch.randn(*s, dtype=dt, device=d)) for s in shapes]
    torch.manual_seed(3)
    ref32 = mk(torch.float32, "cpu")
    ref64 = [torch.nn.Parameter(p.detach().double()) for p in ref32]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref32]
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    o32, o64 = torch.optim.Adam(ref32, **kw), torch.optim.Adam(ref64, **kw)
    opt = FlatAdam(mine, **kw)
    assert all(p.data_ptr() % 512 == 0 for p in mine)           # every tensor keeps an allocator-like alignment
    late, never = 4, 5
    skip = torch.zeros(1, dtype=torch.int32, device=dev)
    for step in range(8):
        if step == 5:
            for o in (o32, o64):
                o.param_groups[0]["lr"] = 2.5e-4
            opt.param_groups[0]["lr"] = 2.5e-4
        gs = [torch.randn(*s) * (0.1 + i) for i, s in enumerate(shapes)]
        for i in range(len(shapes)):
            has = i != never and (i != late or step >= 3)
            ref32[i].grad = gs[i].clone() if has else None
            ref64[i].grad = gs[i].double() if has else None
            # odd steps: a gradient that is a misaligned view (scalar loads in the kernel); even steps: its own tensor
            if has and step % 2 == 1 and gs[i].numel() > 1:
                buf = torch.zeros(gs[i].numel() + 1, device=dev)
                buf[1:].copy_(gs[i].reshape(-1))
                mine[i].grad = buf[1:].view(shapes[i])
            else:
                mine[i].grad = gs[i].to(dev) if has else None
        if step == 6:       # a discarded step: nothing may change, the counters included
            before = [p.detach().clone() for p in mine]
            skip.fill_(1)
            opt.step(skip=skip)
            skip.zero_()
            assert all(torch.equal(a, b) for a, b in zip(before, mine))
        o32.step()
        o64.step()
        opt.step()
    torch.cuda.synchronize()
    for i, (a, b, c) in enumerate(zip(mine, ref32, ref64)):
        noise = float((b.detach().double() - c.detach()).abs().max())
        err = float((a.detach().cpu().double() - c.detach()).abs().max())
        assert err <= 4 * noise + 1e-7, (i, err, noise)
    assert torch.equal(mine[never].detach().cpu(), ref32[never].detach())          # untouched
    sd = opt.state_dict()
    steps = {i: int(s["step"]) for i, s in sd["state"].items()}
    assert never not in steps and steps[late] == 5 and steps[0] == 8, steps
    assert opt.uploads <= 9
    # torch -> flat and flat -> torch through the checkpoint format
    fresh = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt2 = FlatAdam(fresh, **kw)
    opt2.load_state_dict(o32.state_dict())
    t2 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone().cpu()) for p in mine], **kw)
    t2.load_state_dict(sd)
    g = [torch.randn(*s) for s in shapes]
    for i in range(len(shapes)):
        fresh[i].grad = g[i].to(dev) if i != never else None
        t2.param_groups[0]["params"][i].grad = g[i].clone() if i != never else None
        ref32[i].grad = g[i].clone() if i != never else None
    opt2.step(); t2.step(); o32.step()
    for i in range(len(shapes)):
        assert torch.allclose(fresh[i].detach().cpu(), ref32[i].detach(), rtol=0, atol=2e-6), i
        assert torch.allclose(t2.param_groups[0]["params"][i].detach(), ref32[i].detach(), rtol=0, atol=2e-6), i
    assert opt2.param_groups[0]["lr"] == 2.5e-4
