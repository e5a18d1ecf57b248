"""Shared by the CPU (oracle) and GPU (HIP) tests of tests/golden/module_sa_real.npz: the two set-abstraction levels of
the MSG part-segmentation network at their REAL shapes (models/pointnet2_part_seg_msg.py:27-28), see
oracle/make_golden.py:golden_modules_real."""
import numpy as np
import torch

from prifit_amd import synth

CASES = (("sa1", 2048, None, (512, [0.1, 0.2, 0.4], [32, 64, 128], 3, [[32, 32, 64], [64, 64, 128], [64, 96, 128]]), 61, 13),
         ("sa2", 512, 320, (128, [0.4, 0.8], [64, 128], 128 + 128 + 64, [[128, 128, 256], [128, 196, 256]]), 62, 14))
B = 4


def _t(a):
    return torch.from_numpy(np.asarray(a))


def inputs(g, i):
    name, N, C, cfg, tseed, bnseed = CASES[i]
    seed = int(g["seed"])
    xyz = _t(synth.cloud("surface", B, N, seed + 2 * i)).transpose(1, 2).contiguous()
    feat = xyz.clone() if C is None else _t(synth.features(B, N, C, seed + 2 * i + 1)).transpose(1, 2).contiguous()
    start = _t(g[name + "_start"])
    nout = sum(w[-1] for w in cfg[4])
    gout = _t(synth.features(B, cfg[0], nout, seed + 10 + i)).transpose(1, 2)
    return xyz, feat, start, gout


def seeded_module(ctor, i):
    """Same construction order and seeds as the generator => the reference's parameters."""
    name, N, C, cfg, tseed, bnseed = CASES[i]
    torch.manual_seed(tseed)
    m = ctor(*cfg)
    synth.xavier_like_trainer(m)
    synth.perturb_bn(m, bnseed)
    return m


def check(g, i, new_xyz, out, dfeat, grads, running, out_tol=1e-4):
    """Outputs vs the reference's (heads + sums along both axes), gradients in relative L2 against the fixture's measured
    fp32-irreproducibility bound (2 x the largest deviation seen between reference-fp32, oracle-fp32 and oracle-fp64)."""
    name = CASES[i][0]
    assert torch.equal(new_xyz, _t(g[name + "_new_xyz"]))
    ref_head = _t(g[name + "_out_head"])
    scale = max(1.0, float(ref_head.abs().max()))
    torch.testing.assert_close(out[:, :, :8], ref_head, rtol=out_tol, atol=out_tol * scale)
    S, Cc = out.shape[2], out.shape[1]
    torch.testing.assert_close(out.sum(2), _t(g[name + "_out_sum_s"]), rtol=out_tol, atol=out_tol * scale * S ** 0.5)
    torch.testing.assert_close(out.sum(1), _t(g[name + "_out_sum_c"]), rtol=out_tol, atol=out_tol * scale * Cc ** 0.5)
    tol, dtol = float(g[name + "_grad_tol"]), float(g[name + "_dfeat_tol"])
    report = {}
    if dfeat is not None:
        for key, got in (("dfeat_head", dfeat[:, :, :16]), ("dfeat_sum_n", dfeat.sum(2)), ("dfeat_sum_c", dfeat.sum(1))):
            ref = _t(g[name + "_" + key])
            rel = float((got - ref).norm() / ref.norm())
            report[key] = rel
            assert rel <= dtol, (name, key, rel, dtol)
    gmax = max(float(_t(g[k]).abs().max()) for k in g.files if k.startswith(name + "_g_"))
    for k, gr in grads.items():
        ref = _t(g[name + "_g_" + k])
        if k.endswith(".bias") and "conv" in k:
            assert float(gr.abs().max()) <= 1e-3 * gmax, (name, k)     # in front of a train-mode BatchNorm: true value 0
            continue
        rel = float((gr - ref).norm() / ref.norm())
        report[k] = rel
        assert rel <= tol, (name, k, rel, tol)
    rm, rv = running
    torch.testing.assert_close(rm, _t(g[name + "_running_mean_last"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv, _t(g[name + "_running_var_last"]), rtol=1e-4, atol=1e-5)
    return report
