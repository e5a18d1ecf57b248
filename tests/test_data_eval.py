"""Data path and evaluation: device-side pc_normalize / resample / readers vs the numpy
restatement of data_utils/ShapeNetDataLoader.py, and the batched metrics vs the loop restatement of
testing.py:138-240 plus hand-computed cases.  Round 6: the readers, the two augmentation functions and `evaluation` itself
against the REFERENCE's outputs on a seeded synthetic tree (tests/golden/data_readers.npz, eval_metrics.npz, written by
oracle/make_golden.py:golden_data_readers / golden_eval from the reference's own classes and its own `testing.evaluation`).  Pure torch ops: these run on the CPU here and on the GPU in the
gpu-marked variant."""
import json
import os

import numpy as np
import pytest
import torch

import prifit_oracle as orc
from prifit_amd import data as D
from prifit_amd import testing as T


def _clouds(rng, sizes, C=7):
    out = []
    for n in sizes:
        c = rng.normal(size=(n, C)).astype(np.float32) * rng.uniform(0.5, 3.0) + rng.uniform(-2, 2, size=(1, C)).astype(np.float32)
        c[:, -1] = rng.integers(12, 16, size=n)
        out.append(c)
    return out


def _check_normalize(dev):
    rng = np.random.default_rng(0)
    clouds = _clouds(rng, [300, 300, 300])
    pts = torch.from_numpy(np.stack(clouds)).to(dev)
    got = D.pc_normalize(pts[..., :6])
    for b, c in enumerate(clouds):
        np.testing.assert_allclose(got[b, :, :3].cpu().numpy(), orc.pc_normalize_np(c[:, :3]), rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(got[b, :, 3:].cpu().numpy(), c[:, 3:6])      # normals untouched
    # ragged batch through DeviceBatcher: statistics ignore the padding, draws stay inside each cloud
    ragged = _clouds(rng, [257, 300, 123])
    gen = torch.Generator(device=dev).manual_seed(5)
    out, seg, allp, lengths = D.DeviceBatcher(128, dev, gen)([c[:, :3] for c in ragged], [c[:, -1] for c in ragged])
    assert out.shape == (3, 128, 3) and seg.shape == (3, 128) and lengths.tolist() == [257, 300, 123]
    for b, c in enumerate(ragged):
        ref = orc.pc_normalize_np(c[:, :3])
        np.testing.assert_allclose(allp[b, :c.shape[0]].cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
        # every drawn row is a row of the normalised cloud, with its own label
        d = (out[b].cpu()[:, None, :] - torch.from_numpy(ref)[None]).abs().sum(-1)
        j = d.argmin(1)
        assert float(d.min(1)[0].max()) < 1e-5
        np.testing.assert_array_equal(seg[b].cpu().numpy(), c[j.numpy(), -1].astype(np.int64))
    # an explicit choice reproduces numpy's fancy indexing (ShapeNetDataLoader.py:132-135)
    choice = rng.integers(0, 300, size=(3, 64))
    o2, s2, _ = D.resample(pts[..., :3], pts[..., -1].long(), 64, choice=torch.from_numpy(choice))
    for b in range(3):
        np.testing.assert_array_equal(o2[b].cpu().numpy(), clouds[b][choice[b], :3])


def test_normalize_against_reference_fixture(golden):
    g = golden("data_normalize")    # produced by the reference's pc_normalize (oracle/make_golden.py data)
    np.testing.assert_array_equal(orc.pc_normalize_np(g["cloud"].copy()), g["normalized"])
    got = D.pc_normalize(torch.from_numpy(g["cloud"])[None])[0].numpy()
    np.testing.assert_allclose(got, g["normalized"], rtol=1e-5, atol=1e-6)


def test_normalize_resample_cpu():
    _check_normalize(torch.device("cpu"))


@pytest.mark.gpu
def test_normalize_resample_gpu():
    _check_normalize(torch.device("cuda", 0))


def test_dataset_readers(tmp_path):
    rng = np.random.default_rng(1)
    # ShapeNet part layout: synsetoffset2category.txt, train_test_split/*.json, <synset>/<token>.txt
    root = tmp_path / "shapenet"
    (root / "train_test_split").mkdir(parents=True)
    (root / "synsetoffset2category.txt").write_text("Airplane\t02691156\nChair\t03001627\n")
    tokens = {"02691156": ["a1", "a2", "a3"], "03001627": ["c1", "c2"]}
    split = {"train": ["a1", "c1"], "val": ["a2"], "test": ["a3", "c2"]}
    raw = {}
    for syn, toks in tokens.items():
        (root / syn).mkdir()
        for t in toks:
            arr = np.concatenate([rng.normal(size=(50, 6)), rng.integers(0, 4, size=(50, 1))], 1)
            np.savetxt(root / syn / (t + ".txt"), arr)
            raw[t] = arr.astype(np.float32)
    for s, toks in split.items():
        names = ["shape_data/%s/%s" % (syn, t) for syn, ts in tokens.items() for t in ts if t in toks]
        (root / "train_test_split" / ("shuffled_%s_file_list.json" % s)).write_text(json.dumps(names))
    ds = D.PartNormalDataset(str(root), npoints=32, split="trainval", rng=np.random.default_rng(7))
    assert len(ds) == 3 and ds.classes == {"Airplane": 0, "Chair": 1}
    pts, cls, seg = ds[0]
    assert pts.shape == (32, 3) and cls.tolist() == [0] and seg.shape == (32,)
    ref = orc.pc_normalize_np(raw["a1"][:, :3])
    assert np.abs(pts[:, None, :] - ref[None]).sum(-1).min(1).max() < 1e-5      # rows of the normalised cloud
    test = D.PartNormalDataset(str(root), npoints=16, split="test", normal_channel=True, raw=True)
    assert len(test) == 2 and test[1][0].shape == (50, 6)
    with pytest.raises(ValueError):
        D.PartNormalDataset(str(root), split="nope")
    # ACD layout: <root>/<subfolder>/<token>.npy
    acd = tmp_path / "acd"
    for sub in ("chairs", "planes"):
        (acd / sub).mkdir(parents=True)
        for i in range(2):
            np.save(acd / sub / ("%s%d.npy" % (sub, i)), np.concatenate([rng.normal(size=(80, 3)), rng.integers(0, 9, size=(80, 1))], 1))
    ad = D.ACDSelfSupDataset(str(acd), npoints=40, exclude_fns=["x/planes1.npy"], rng=np.random.default_rng(3))
    assert len(ad) == 3
    # the trainer passes the labeled datasets' '.txt' paths (train_partseg_shapenet.py:194-210): tokens are compared
    # without their extension (ShapeNetDataLoader.py:305-311), and both datasets expose `.meta` for that purpose
    labeled = ["/some/where/chairs0.txt", "/else/planes1.txt"]
    ad2 = D.ACDSelfSupDataset(str(acd), npoints=40, exclude_fns=labeled)
    assert sorted(os.path.basename(f) for _, f in ad2.datapath) == ["chairs1.npy", "planes0.npy"]
    assert sorted(ad2.meta) == ["chairs", "planes"] and sum(len(v) for v in ad2.meta.values()) == 2
    assert sorted(ds.meta) == ["Airplane", "Chair"] and all(f.endswith(".txt") for v in ds.meta.values() for f in v)
    assert [f for v in ds.meta.values() for f in v] == [f for _, f in ds.datapath]
    p, allp, cls, seg = ad[0]
    assert p.shape == (40, 3) and allp.shape == (80, 3) and abs(np.sqrt((allp ** 2).sum(1)).max() - 1.0) < 1e-5


def _check_metrics(dev):
    rng = np.random.default_rng(2)
    batches = []
    for _ in range(3):
        cats = rng.integers(0, 16, size=5)
        tgt = np.stack([rng.choice(T.seg_classes[T.classes[c]], size=200) for c in cats]).astype(np.int64)
        logits = rng.normal(size=(5, 200, 50)).astype(np.float32)
        logits[np.arange(5)[:, None], np.arange(200)[None], tgt] += 1.5     # mostly right
        batches.append((logits, tgt))
    ev = T.SegmentationEvaluator()
    for lg, tg in batches:
        ev.update(torch.from_numpy(lg).to(dev), torch.from_numpy(tg).to(dev))
    got, ref = ev.compute(), orc.eval_metrics_loops(batches)
    for k in ("accuracy", "class_avg_iou", "instance_avg_iou"):
        assert abs(got[k] - ref[k]) < 1e-12, k
    assert (np.isnan(got["class_avg_accuracy"]) and np.isnan(ref["class_avg_accuracy"])) or \
        abs(got["class_avg_accuracy"] - ref["class_avg_accuracy"]) < 1e-12
    for c, v in ref["category_iou"].items():
        assert (np.isnan(v) and np.isnan(got["category_iou"][c])) or abs(v - got["category_iou"][c]) < 1e-12
    # hand-computed: one Bag shape (parts 4, 5), 4 points, target 4 4 5 5; logits prefer part 4 on three points ->
    # prediction 4 4 4 5: IoU(4) = 2/3, IoU(5) = 1/2, shape IoU = 7/12, accuracy 3/4; a Cap shape (parts 6, 7)
    # with only part 6 present and predicted: IoU(6) = 1, IoU(7) = 1 (absent in both)
    lg = np.full((2, 4, 50), -5.0, dtype=np.float32)
    lg[0, :, 20] = 9.0                    # a foreign part with a high score must be ignored
    lg[0, [0, 1, 2], 4] = 1.0
    lg[0, 3, 5] = 1.0
    lg[1, :, 6] = 1.0
    tg = np.array([[4, 4, 5, 5], [6, 6, 6, 6]])
    ev = T.SegmentationEvaluator()
    pred = ev.update(torch.from_numpy(lg).to(dev), torch.from_numpy(tg).to(dev))
    assert pred.tolist() == [[4, 4, 4, 5], [6, 6, 6, 6]]
    m = ev.compute()
    assert abs(m["category_iou"]["Bag"] - 7.0 / 12.0) < 1e-12 and m["category_iou"]["Cap"] == 1.0
    assert abs(m["instance_avg_iou"] - (7.0 / 12.0 + 1.0) / 2) < 1e-12 and abs(m["accuracy"] - 7.0 / 8.0) < 1e-12
    ref = orc.eval_metrics_loops([(lg, tg)])
    assert abs(ref["instance_avg_iou"] - m["instance_avg_iou"]) < 1e-12


def test_metrics_cpu():
    _check_metrics(torch.device("cpu"))


@pytest.mark.gpu
def test_metrics_and_evaluation_loop_gpu(hiplib):
    _check_metrics(torch.device("cuda", 0))
    # the evaluation loop around the MSG model (random weights): shapes, keys, eval/train mode restored
    from prifit_amd.models import pointnet2_part_seg_msg as M
    torch.manual_seed(0)
    net = M.get_model(50, normal_channel=False).cuda().train()
    rng = np.random.default_rng(4)
    loader = []
    for _ in range(2):
        cats = rng.integers(0, 16, size=2)
        pts = rng.uniform(-1, 1, size=(2, 512, 3)).astype(np.float32)
        tgt = np.stack([rng.choice(T.seg_classes[T.classes[c]], size=512) for c in cats])
        loader.append((pts, cats.reshape(2, 1), tgt))
    metrics = {"best_class_avg_miou": -1.0}
    out = T.evaluate_loader(net, loader, metrics=metrics, epoch=3)
    assert net.training and 0.0 <= out["accuracy"] <= 1.0 and 0.0 <= out["instance_avg_iou"] <= 1.0
    assert metrics["best_epoch"] == 4 and metrics["best_class_avg_miou"] == out["class_avg_iou"]


# ----------------------------------------------------------------------------------------------------------------------
# against the reference's outputs (oracle/make_golden.py data): same seeded trees, same replayed np.random / random states
# ----------------------------------------------------------------------------------------------------------------------
TREE_SEED, ACD_SEED, NPOINT, ITEM_SEED = 21, 22, 48, 1000     # oracle/make_golden.py


def _rel(paths, root):
    return [os.path.relpath(p, root) for p in paths]


def test_dataset_readers_match_reference_golden(golden, tmp_path):
    """PartNormalDataset (three configurations + the few-shot draw), SelfSupPartNormalDataset and ACDSelfSupDataset with the
    trainer's overlap removal: file lists, class ids and every item (points, label, resampled segmentation) bit for bit."""
    import random
    from prifit_amd import synth
    g = golden("data_readers")
    root, acd = str(tmp_path / "shapenet"), str(tmp_path / "acd")
    synth.write_partseg_tree(root, TREE_SEED)
    for tag, kw in (("trainval", dict(split="trainval", normal_channel=False)), ("test_n", dict(split="test", normal_channel=True)),
                    ("train_car_chair", dict(split="train", normal_channel=False, class_choice=["Car", "Chair"]))):
        ds = D.PartNormalDataset(root=root, npoints=NPOINT, **kw)
        assert len(ds) == int(g["pn_%s_n" % tag])
        assert _rel([fn for _, fn in ds.datapath], root) == g["pn_%s_paths" % tag].tolist()
        assert [c for c, _ in ds.datapath] == g["pn_%s_cats" % tag].tolist()
        assert ["%s=%d" % kv for kv in sorted(ds.classes.items())] == g["pn_%s_classes" % tag].tolist()
        for i in range(len(ds)):
            np.random.seed(ITEM_SEED + i)
            pts, cls, seg = ds[i]
            np.testing.assert_array_equal(pts, g["pn_%s_pts_%d" % (tag, i)])
            np.testing.assert_array_equal(cls, g["pn_%s_cls_%d" % (tag, i)])
            np.testing.assert_array_equal(seg, g["pn_%s_seg_%d" % (tag, i)])
            assert pts.dtype == np.float32 and cls.dtype == np.int32 and seg.dtype == np.int32
    random.seed(5)
    ds = D.PartNormalDataset(root=root, npoints=NPOINT, split="trainval", k_shot=1)
    assert _rel([fn for _, fn in ds.datapath], root) == g["pn_kshot_paths"].tolist()
    train = D.PartNormalDataset(root=root, npoints=NPOINT, split="train", k_shot=-1)
    test = D.PartNormalDataset(root=root, npoints=NPOINT, split="test")
    labeled = [f for v in test.meta.values() for f in v] + [f for v in train.meta.values() for f in v]
    ss = D.SelfSupPartNormalDataset(root=root, npoints=NPOINT, split="trainval", labeled_fns=labeled)
    assert len(ss) == int(g["ss_n"]) and _rel([fn for _, fn in ss.datapath], root) == g["ss_paths"].tolist()
    for i in range(len(ss)):
        np.random.seed(ITEM_SEED + i)
        pts, cls, seg = ss[i]
        np.testing.assert_array_equal(pts, g["ss_pts_%d" % i])
        np.testing.assert_array_equal(cls, g["ss_cls_%d" % i])
        np.testing.assert_array_equal(seg, g["ss_seg_%d" % i])
    with pytest.raises(TypeError):
        D.SelfSupPartNormalDataset(root=root)            # labeled_fns is required (the reference iterates over it, :158)
    # ACD: items are compared per file token (the reference lists sub-folders in os.listdir order, this package sorted)
    synth.write_acd_tree(acd, ACD_SEED, overlap_tokens=["chair_t0", "lamp_t2"])
    ad = D.ACDSelfSupDataset(root=acd, npoints=NPOINT, exclude_fns=labeled, prefetch=False)
    toks = [os.path.splitext(os.path.basename(fn))[0] for _, fn in ad.datapath]
    ref_toks = g["acd_tokens"].tolist()
    assert len(ad) == int(g["acd_n"]) and sorted(toks) == sorted(ref_toks)
    assert dict(zip(toks, [c for c, _ in ad.datapath])) == dict(zip(ref_toks, g["acd_cats"].tolist()))
    for i, tok in enumerate(toks):
        np.random.seed(ITEM_SEED + ref_toks.index(tok))
        pts, cham, cls, seg = ad[i]
        np.testing.assert_array_equal(pts, g["acd_pts_" + tok])
        np.testing.assert_array_equal(cham, g["acd_cham_" + tok])
        np.testing.assert_array_equal(seg, g["acd_seg_" + tok])
        assert cls.tolist() == [ad.classes[ad.datapath[i][0]]]
    # prefetch=True: the items are drawn once, at construction (:340-366), in dataset order from one np.random stream
    np.random.seed(9)
    pre = D.ACDSelfSupDataset(root=acd, npoints=NPOINT, exclude_fns=labeled, prefetch=True)
    np.random.seed(9)
    lazy = D.ACDSelfSupDataset(root=acd, npoints=NPOINT, exclude_fns=labeled)
    for i in range(len(pre)):
        for a, b in zip(pre[i], lazy[i]):
            np.testing.assert_array_equal(a, b)
        assert pre[i][0] is pre[i][0]                    # stored, not re-drawn


def test_provider_scale_shift_match_reference_golden(golden):
    """provider.random_scale_point_cloud / shift_point_cloud as train_partseg_shapenet.py:372-373 calls them, one seeded
    np.random state: the reference's batch bit for bit (numpy path); the torch path follows the same law on the tensor's device."""
    from prifit_amd import provider
    g = golden("data_readers")
    aug = g["aug_in"].copy()
    np.random.seed(77)
    aug[:, :, 0:3] = provider.random_scale_point_cloud(aug[:, :, 0:3])
    aug[:, :, 0:3] = provider.shift_point_cloud(aug[:, :, 0:3])
    np.testing.assert_array_equal(aug, g["aug_out"])
    x = torch.from_numpy(g["aug_in"]).clone()
    gen = torch.Generator().manual_seed(1)
    y = provider.shift_point_cloud(provider.random_scale_point_cloud(x.clone(), generator=gen), generator=gen)
    # one scale in [0.8, 1.25] and one shift in [-0.1, 0.1]^3 per cloud: solve them back from two points of every cloud
    sc = (y[:, 1] - y[:, 0]) / (x[:, 1] - x[:, 0])
    assert torch.allclose(sc, sc[:, :1].expand(-1, 3), atol=1e-4) and bool(((sc >= 0.8 - 1e-4) & (sc <= 1.25 + 1e-4)).all())
    sh = y[:, 0] - sc * x[:, 0]
    assert bool((sh.abs() <= 0.1 + 1e-4).all()) and torch.allclose(y, x * sc[:, None, :1] + sh[:, None], atol=1e-5)


def _eval_args(**over):
    """The fields of args_parser.py's namespace that testing.evaluation reads (testing.py:55-139); oracle/make_golden.py:eval_args."""
    import argparse
    a = dict(gpu=None, cudnn_off=False, eval_split="test", npoint=NPOINT, normal=False, batch_size=5, num_classes=16, num_parts=50,
             seed=3, pretrained_model=None, model="models.pointnet2_part_seg_msg", category=True, if_cuboid=False, quantile=0.05,
             msc_iterations=10, max_num_clusters=25, alpha=1.0, beta=1.0, embed=False, reconstruct=False, dgcnn_k=20)
    a.update(over)
    return argparse.Namespace(**a)


def _check_evaluation_against_reference(golden, tmp_path, monkeypatch, dev):
    from prifit_amd import synth
    g = golden("eval_metrics")
    synth.write_partseg_tree(str(tmp_path / T.DATA_ROOT), TREE_SEED)
    monkeypatch.chdir(tmp_path)
    net = orc.StubSegClassifier(50, seed=4).to(dev)
    metrics = {"best_class_avg_miou": -1.0, "best_acc": 0.0, "best_epoch": 0, "best_instance_avg_miou": 0.0, "best_chamfer_loss": 1e9}
    np.random.seed(123)
    ret = T.evaluation(_eval_args(), 6, net, metrics)           # the reference's call, train_partseg_shapenet.py:487
    assert ret is metrics and ret["best_epoch"] == int(g["best_epoch"]) == 7
    for k, gk in (("best_acc", "accuracy"), ("best_class_avg_miou", "class_avg_iou"), ("best_instance_avg_miou", "instance_avg_iou")):
        assert abs(ret[k] - float(g[gk])) < 1e-12, (k, ret[k], float(g[gk]))
    assert abs(ret["best_chamfer_loss"] - float(g["chamfer_loss"])) < 1e-6
    last = T.evaluation.last
    for name, v in zip(g["category_names"].tolist(), g["category_iou"].tolist()):
        assert abs(last["category_iou"][name] - v) < 1e-6, name         # (the reference prints them with %f)
    # the keywords the reference's loop hands to the classifier (testing.py:139): every one this package's models accept
    assert len(net.calls) == int(g["n_batches"])
    sent = set(net.calls[0])
    assert {"include_convex_loss", "evaluation", "quantile", "msc_iterations", "max_num_clusters", "alpha", "beta", "if_cuboid",
            "embed", "seed"} <= sent <= set(g["forward_kwargs"].tolist())
    # a worse class-average mIoU leaves the running best alone (testing.py:241-247)
    keep = dict(ret, best_class_avg_miou=2.0)
    np.random.seed(123)
    assert T.evaluation(_eval_args(), 9, orc.StubSegClassifier(50, seed=4).to(dev), dict(keep)) == keep
    return g


def test_evaluation_matches_reference_golden_cpu(golden, tmp_path, monkeypatch):
    g = _check_evaluation_against_reference(golden, tmp_path, monkeypatch, torch.device("cpu"))
    # ... and the ORACLE's metric loops (oracle/prifit_oracle.py:eval_metrics_loops) on the same batches: pinned by the
    # reference's numbers, no longer by hand-computed cases alone
    ds = D.PartNormalDataset(root=T.DATA_ROOT, npoints=NPOINT, split="test")
    net = orc.StubSegClassifier(50, seed=4)
    np.random.seed(123)
    batches = []
    for lo in range(0, len(ds), 5):
        items = [ds[i] for i in range(lo, min(lo + 5, len(ds)))]
        pts = torch.from_numpy(np.stack([it[0] for it in items])).transpose(2, 1)
        lab = torch.from_numpy(np.stack([it[1] for it in items])).long()
        seg = net(pts, T.to_categorical(lab, 16))[0]
        batches.append((seg.numpy(), np.stack([it[2] for it in items])))
    m = orc.eval_metrics_loops(batches)
    for k in ("accuracy", "class_avg_iou", "instance_avg_iou"):
        assert abs(float(m[k]) - float(g[k])) < 1e-12, k


@pytest.mark.gpu
def test_evaluation_matches_reference_golden_gpu(golden, tmp_path, monkeypatch):
    _check_evaluation_against_reference(golden, tmp_path, monkeypatch, torch.device("cuda", 0))
