"""Seeded synthetic inputs (numpy default_rng) shared by the oracle harness, the tests and bench.py (lives in the package, not under oracle/:
the product entry points never import from oracle/).

TEST/BENCH INFRASTRUCTURE: pure data generation, no reference code and no model code.
Distributions follow SURVEY.md section 8(d):
  cube     xyz ~ U[-1,1]^3                      (BASELINE.json wording: synthetic clouds)
  surface  points on a random ellipsoid surface, centred and scaled to the unit ball
           (what ShapeNet clouds look like after data_utils/ShapeNetDataLoader.py:17-22)
  blobs    8 Gaussian blobs (sigma .08) with centres ~ U[-.8,.8]^3: gives non-trivial clusters
"""
import numpy as np


def cloud(kind, B, N, seed):
    rng = np.random.default_rng(seed)
    if kind == "cube":
        return rng.uniform(-1.0, 1.0, size=(B, N, 3)).astype(np.float32)
    if kind == "surface":
        out = np.empty((B, N, 3), dtype=np.float32)
        for b in range(B):
            axes = rng.uniform(0.3, 1.0, size=3)
            v = rng.normal(size=(N, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            p = v * axes
            p -= p.mean(axis=0, keepdims=True)
            p /= np.sqrt((p ** 2).sum(axis=1)).max()
            out[b] = p.astype(np.float32)
        return out
    if kind == "blobs":
        out = np.empty((B, N, 3), dtype=np.float32)
        for b in range(B):
            centres = rng.uniform(-0.8, 0.8, size=(8, 3))
            lab = rng.integers(0, 8, size=N)
            out[b] = (centres[lab] + 0.08 * rng.normal(size=(N, 3))).astype(np.float32)
        return out
    raise ValueError(kind)


def blobs_with_labels(B, N, seed, K=8, sigma=0.08):
    """Blob cloud plus the generating label of every point (for prototype embeddings)."""
    rng = np.random.default_rng(seed)
    pts = np.empty((B, N, 3), dtype=np.float32)
    labs = np.empty((B, N), dtype=np.int64)
    for b in range(B):
        centres = rng.uniform(-0.8, 0.8, size=(K, 3))
        lab = rng.integers(0, K, size=N)
        pts[b] = (centres[lab] + sigma * rng.normal(size=(N, 3))).astype(np.float32)
        labs[b] = lab
    return pts, labs


def prototype_embedding(labels, D, seed, K=8, noise=0.01):
    """normalize(proto[label] + noise * N(0,1)), proto = normalised N(0,1)[K, D]."""
    rng = np.random.default_rng(seed)
    proto = rng.normal(size=(K, D))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    e = proto[labels] + noise * rng.normal(size=labels.shape + (D,))
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    return e.astype(np.float32)


def fps_start(B, N, seed):
    return np.random.default_rng(seed + 7919).integers(0, N, size=B).astype(np.int64)


def features(B, N, C, seed):
    return np.random.default_rng(seed + 104729).normal(size=(B, N, C)).astype(np.float32)


def labels(B, N, num_classes, seed):
    return np.random.default_rng(seed + 1299709).integers(0, num_classes, size=(B, N)).astype(np.int64)


def uniform01(shape, seed):
    return np.random.default_rng(seed + 15485863).uniform(0.0, 1.0, size=shape).astype(np.float32)


def perturb_bn(module, seed):
    """Give every BatchNorm non-trivial affine parameters (incl. negative gammas) and running stats,
    so that parity tests exercise the gamma<0 pooling branch and the running-stat update."""
    import torch

    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def xavier_like_trainer(module):
    """train_partseg_shapenet.py:240-247: xavier_normal_ on Conv2d/Linear weights, zero bias."""
    import torch

    for m in module.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            torch.nn.init.xavier_normal_(m.weight.data)
            torch.nn.init.constant_(m.bias.data, 0.0)


def ellipsoid_scene(B, seed, n_per=500, D=32):
    """The stand-alone fitting demo's data (fitting.py:26-31 / src/ellipsoid_fitting.py:144-193, with the build's
    Fibonacci table in place of trimesh's sampler): per shape 3 ellipsoids with integer semi-axes in [2, 20), a random
    rotation about z and a random centre in [0, max axis)^3, `n_per` surface points each, and D-dimensional one-hot
    "embeddings".  Returns points [B,3*n_per,3] f32, X [B,3*n_per,D] f32, abc [B,3,3], centres [B,3,3]."""
    rng = np.random.default_rng(seed)
    j = np.arange(n_per, dtype=np.float64)
    z = 1.0 - (2.0 * j + 1.0) / n_per
    lon = 2.0 * np.pi * np.modf(j * 0.6180339887498949)[0]
    sv = np.sqrt(1.0 - z * z)
    unit = np.stack([np.cos(lon) * sv, np.sin(lon) * sv, z], 1)
    pts = np.zeros((B, 3 * n_per, 3), np.float32)
    X = np.zeros((B, 3 * n_per, D), np.float32)
    abc = np.zeros((B, 3, 3))
    ctr = np.zeros((B, 3, 3))
    for b in range(B):
        for k in range(3):
            axes = rng.integers(2, 20, size=3).astype(np.float64)
            th = rng.uniform(0, 2 * np.pi)
            Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
            c = rng.uniform(0, 1, size=3) * axes.max()
            pts[b, k * n_per:(k + 1) * n_per] = ((unit * axes) @ Rz + c).astype(np.float32)
            X[b, k * n_per:(k + 1) * n_per, k] = 1.0
            abc[b, k], ctr[b, k] = axes, c
    return pts, X, abc, ctr


def part_labels(points, K=8, seed=0):
    """A spatial partition of every cloud into K parts (what a trained part embedding encodes): the Voronoi cells of K
    of its own points, chosen with a seeded permutation.  points [B,N,3] -> int64 [B,N]."""
    rng = np.random.default_rng(seed + 32452843)
    B, N, _ = points.shape
    out = np.empty((B, N), dtype=np.int64)
    for b in range(B):
        anchors = points[b][rng.permutation(N)[:K]]
        out[b] = ((points[b][:, None, :] - anchors[None]) ** 2).sum(-1).argmin(1)
    return out


def equal_part_labels(points, nx, ny):
    """A spatial partition of every cloud into nx * ny parts of (nearly) EQUAL size: nx slabs along x, each cut into ny
    along y.  Equal sizes keep the mean-shift bandwidth statistic (the mean k-th neighbour distance) inside the parts for
    every point, so the number of modes is the number of parts.  points [B,N,3] -> int64 [B,N]."""
    B, N, _ = points.shape
    out = np.empty((B, N), dtype=np.int64)
    for b in range(B):
        ox = np.argsort(points[b][:, 0], kind="stable")
        for i, slab in enumerate(np.array_split(ox, nx)):
            oy = slab[np.argsort(points[b][slab, 1], kind="stable")]
            for j, cell in enumerate(np.array_split(oy, ny)):
                out[b][cell] = i * ny + j
    return out


def part_embedding_offset(labels, D=128, seed=0, K=8, noise=0.03, scale=30.0):
    """What a TRAINED embedding head adds to an untrained one, as an explicit input: `scale` x (a random unit prototype
    per part label + `noise` x N(0,1), normalised).  A seeded untrained PointNet++ maps every point of a shape to nearly
    the same embedding direction: mean-shift finds ONE cluster per shape and the fit path downstream of it (nms, membership,
    ellipsoid fit, SDF, sampler) runs at 1/8 .. 1/25 of its training-time work (the reference's regime is up to 25
    clusters, README.md:62).  `convex_loss(..., embedding_offset=...)` adds this tensor to the network's embedding before
    the normalisation -- the head's own output (norm ~8, one common direction) then only tilts the K prototypes, the
    clusters are the parts, memberships are soft (noise 0.03), and the gradient still flows through head and backbone.
    labels int64 [B,N] -> float32 [B,N,D]."""
    return (scale * prototype_embedding(np.asarray(labels), D, seed + 2, K=K, noise=noise)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------------
# Synthetic dataset TREES in the reference's two on-disk formats (data_utils/ShapeNetDataLoader.py:24-140, :265-412): the
# same seeded files for oracle/make_golden.py (which runs the reference's dataset classes and evaluation on them) and for the
# tests (which run prifit_amd/data.py and prifit_amd/testing.py on them and compare with the stored outputs).
# ----------------------------------------------------------------------------------------------------------------------
PARTSEG_CATEGORIES = [("Airplane", "02691156", [0, 1, 2, 3]), ("Bag", "02773838", [4, 5]), ("Cap", "02954340", [6, 7]),
                      ("Car", "02958343", [8, 9, 10, 11]), ("Chair", "03001627", [12, 13, 14, 15]),
                      ("Earphone", "03261776", [16, 17, 18]), ("Guitar", "03467517", [19, 20, 21]), ("Knife", "03624134", [22, 23]),
                      ("Lamp", "03636649", [24, 25, 26, 27]), ("Laptop", "03642806", [28, 29]),
                      ("Motorbike", "03790512", [30, 31, 32, 33, 34, 35]), ("Mug", "03797390", [36, 37]),
                      ("Pistol", "03948459", [38, 39, 40]), ("Rocket", "04099429", [41, 42, 43]),
                      ("Skateboard", "04225987", [44, 45, 46]), ("Table", "04379243", [47, 48, 49])]


def write_partseg_tree(root, seed=0):
    """ShapeNet part-annotation layout under `root`: synsetoffset2category.txt, train_test_split/shuffled_{train,val,test}
    _file_list.json, <synset>/<token>.txt with rows `x y z nx ny nz label`.  All 16 categories (the reference's class-average
    mIoU is nan unless every category has a test shape, testing.py:226-234), per category tokens t0 (train), t1 (val), t2 (test)
    and for every third category t3 (test); 24..63 points per file, labels drawn from the category's parts (the first row's
    label decides the category at evaluation time, testing.py:142).  Returns {token: array [n, 7] float64 as written}."""
    import json
    import os
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "train_test_split"), exist_ok=True)
    with open(os.path.join(root, "synsetoffset2category.txt"), "w") as f:
        for name, syn, _ in PARTSEG_CATEGORIES:
            f.write("%s\t%s\n" % (name, syn))
    split = {"train": [], "val": [], "test": []}
    written = {}
    for ci, (name, syn, parts) in enumerate(PARTSEG_CATEGORIES):
        os.makedirs(os.path.join(root, syn), exist_ok=True)
        toks = [("t0", "train"), ("t1", "val"), ("t2", "test")] + ([("t3", "test")] if ci % 3 == 0 else [])
        for t, s in toks:
            token = "%s_%s" % (name.lower(), t)
            n = int(rng.integers(24, 64))
            xyz = rng.normal(size=(n, 3)) * rng.uniform(0.5, 2.0, size=(1, 3)) + rng.uniform(-1.5, 1.5, size=(1, 3))
            nrm = rng.normal(size=(n, 3))
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            lab = rng.choice(parts, size=n).astype(np.float64)
            arr = np.concatenate([xyz, nrm, lab[:, None]], 1)
            np.savetxt(os.path.join(root, syn, token + ".txt"), arr)
            written[token] = arr
            split[s].append("shape_data/%s/%s" % (syn, token))
    for s, names in split.items():
        with open(os.path.join(root, "train_test_split", "shuffled_%s_file_list.json" % s), "w") as f:
            json.dump(names, f)
    return written


def write_acd_tree(root, seed=0, overlap_tokens=()):
    """Self-supervised layout under `root`: <sub-folder>/<token>.npy with rows `x y z label` (ACD component ids).  Two
    sub-folders x three files, 40..79 points each, plus one file per `overlap_tokens` entry in the first sub-folder (tokens
    of labeled shapes: the trainer's overlap removal must drop them, train_partseg_shapenet.py:190-210).
    Returns {token: array [n, 4] float64}."""
    import os
    rng = np.random.default_rng(seed)
    written = {}
    for si, sub in enumerate(("acd_a", "acd_b")):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        toks = ["%s_%d" % (sub, i) for i in range(3)] + (list(overlap_tokens) if si == 0 else [])
        for token in toks:
            n = int(rng.integers(40, 80))
            xyz = rng.normal(size=(n, 3)) * rng.uniform(0.5, 2.0, size=(1, 3)) + rng.uniform(-1.0, 1.0, size=(1, 3))
            lab = rng.integers(0, 9, size=n).astype(np.float64)
            arr = np.concatenate([xyz, lab[:, None]], 1)
            np.save(os.path.join(root, sub, token + ".npy"), arr)
            written[token] = arr
    return written
