"""Data path and evaluation: device-side pc_normalize / resample / readers vs the numpy
restatement of data_utils/ShapeNetDataLoader.py, and the batched metrics vs the loop restatement of
testing.py:138-240 plus hand-computed cases.  Pure torch ops: these run on the CPU here and on the GPU in the
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
    out = T.evaluation(net, loader, metrics=metrics, epoch=3)
    assert net.training and 0.0 <= out["accuracy"] <= 1.0 and 0.0 <= out["instance_avg_iou"] <= 1.0
    assert metrics["best_epoch"] == 4 and metrics["best_class_avg_miou"] == out["class_avg_iou"]
