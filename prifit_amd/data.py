"""Data path either side of the hot path: the reference's per-sample numpy work moved to
batched device ops, plus readers for its two on-disk formats.

    pc_normalize(points)                      data_utils/ShapeNetDataLoader.py:17-22, batched [B,N,3+]
    resample(points, seg, npoints, gen)       :132-135 / :404-407  np.random.choice(len, npoints, replace=True)
    random_scale_shift(points)                provider.py:278-303 (lives in train_step.py, re-exported here)
    PartNormalDataset / SelfSupPartNormalDataset / ACDSelfSupDataset
                                              :24-140 / :149-262 / :265-412   same constructor arguments, folder layout,
                                              split files and __getitem__ contract (numpy out, the reference's draws from
                                              `np.random` / `random`), so the trainer's DataLoader code
                                              (train_partseg_shapenet.py:170-213) keeps working; pinned against the
                                              reference's own classes on a synthetic tree (tests/golden/data_readers.npz)
    DeviceBatcher                             takes the RAW clouds of a batch (ragged point counts) and does
                                              normalise + resample on the device in one go

The datasets only parse directories and read files (host work by nature); everything numeric is a torch op that
runs on whatever device the tensors live on.
"""
import json
import math
import os
import random

import numpy as np
import torch

from .train_step import random_scale_shift  # noqa: F401  (re-export)

SEG_CLASSES = {'Earphone': [16, 17, 18], 'Motorbike': [30, 31, 32, 33, 34, 35], 'Rocket': [41, 42, 43],
               'Car': [8, 9, 10, 11], 'Laptop': [28, 29], 'Cap': [6, 7], 'Skateboard': [44, 45, 46],
               'Mug': [36, 37], 'Guitar': [19, 20, 21], 'Bag': [4, 5], 'Lamp': [24, 25, 26, 27],
               'Table': [47, 48, 49], 'Airplane': [0, 1, 2, 3], 'Pistol': [38, 39, 40],
               'Chair': [12, 13, 14, 15], 'Knife': [22, 23]}   # ShapeNetDataLoader.py:97-102


def pc_normalize(points, lengths=None):
    """Centre every cloud on its centroid and scale it into the unit ball (ShapeNetDataLoader.py:17-22).
    points [B,N,C>=3]: only the first three channels change.  lengths [B] (optional): number of real rows per
    cloud when the batch is padded; padding rows are ignored by the statistics and left as they are."""
    xyz = points[..., :3]
    B, N, _ = xyz.shape
    if lengths is None:
        centroid = xyz.mean(dim=1, keepdim=True)
        centred = xyz - centroid
        m = centred.pow(2).sum(-1).sqrt().amax(dim=1).view(B, 1, 1)
        out_xyz = centred / m
    else:
        live = (torch.arange(N, device=xyz.device).view(1, N) < lengths.view(B, 1)).unsqueeze(-1)
        cnt = lengths.view(B, 1, 1).to(xyz.dtype)
        centroid = (xyz * live).sum(dim=1, keepdim=True) / cnt
        centred = xyz - centroid
        m = (centred.pow(2).sum(-1).sqrt() * live.squeeze(-1)).amax(dim=1).view(B, 1, 1)
        out_xyz = torch.where(live, centred / m, xyz)
    if points.shape[-1] == 3:
        return out_xyz
    return torch.cat([out_xyz, points[..., 3:]], dim=-1)


def resample(points, seg, npoints, generator=None, lengths=None, choice=None):
    """Draw `npoints` rows per cloud with replacement (ShapeNetDataLoader.py:132-135).  points [B,N,C], seg [B,N]
    (or None).  `choice` [B,npoints] overrides the random draw (tests, reproducing a numpy draw)."""
    B, N, _ = points.shape
    dev = points.device
    if choice is None:
        u = torch.rand(B, npoints, device=dev, generator=generator)
        hi = (lengths if lengths is not None else torch.full((B,), N, device=dev)).view(B, 1).to(u.dtype)
        choice = (u * hi).long().clamp_(max=N - 1)
        if lengths is not None:
            choice = torch.minimum(choice, (lengths.view(B, 1) - 1).long())
    idx = choice.to(dev).long()
    out = torch.gather(points, 1, idx.unsqueeze(-1).expand(-1, -1, points.shape[-1]))
    return out, (torch.gather(seg, 1, idx) if seg is not None else None), idx


class DeviceBatcher:
    """Raw clouds of one batch (list of [n_i, C] arrays, n_i ragged) -> padded device tensor -> pc_normalize ->
    resample, i.e. what PartNormalDataset.__getitem__ does per sample on the host, done once per batch on the GPU.
    Returns (points [B,npoints,C], seg [B,npoints] or None, all_points [B,Nmax,C], lengths [B])."""

    def __init__(self, npoints, device, generator=None):
        self.npoints, self.device, self.generator = npoints, device, generator

    def __call__(self, clouds, segs=None):
        B = len(clouds)
        C = clouds[0].shape[1]
        lengths = torch.tensor([c.shape[0] for c in clouds])
        nmax = int(lengths.max())
        host = torch.zeros(B, nmax, C, dtype=torch.float32)
        host_seg = torch.zeros(B, nmax, dtype=torch.int64) if segs is not None else None
        for b, c in enumerate(clouds):
            host[b, :c.shape[0]] = torch.as_tensor(np.asarray(c, dtype=np.float32))
            if segs is not None:
                host_seg[b, :c.shape[0]] = torch.as_tensor(np.asarray(segs[b]).astype(np.int64))
        pts = host.to(self.device, non_blocking=True)
        lengths = lengths.to(self.device)
        seg = host_seg.to(self.device, non_blocking=True) if segs is not None else None
        allp = pc_normalize(pts, lengths)
        out, out_seg, _ = resample(allp, seg, self.npoints, self.generator, lengths)
        return out, out_seg, allp, lengths


# --------------------------------------------------------------------------------------------------
# readers (host): same folder layout and item contract as the reference's datasets
# --------------------------------------------------------------------------------------------------
def _normalize_np(pc):
    """numpy twin of pc_normalize for the per-item path of the datasets (ShapeNetDataLoader.py:17-22)."""
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def _few_shot(fns, k, rng):
    """`random.sample(fns, k)` (ShapeNetDataLoader.py:77-79, :208-210): Python's `random` module, as upstream, unless the
    caller passed a numpy generator."""
    if rng is None:
        return random.sample(fns, k)
    return list(rng.choice(fns, k, replace=False))


class PartNormalDataset(torch.utils.data.Dataset):
    """ShapeNet part annotation benchmark, `<synset>/<token>.txt` rows `x y z nx ny nz label`
    (ShapeNetDataLoader.py:24-140).  raw=True returns the un-normalised, un-resampled arrays for DeviceBatcher.
    rng: a numpy Generator for the resampling / few-shot draws (default: the global `np.random` / `random` state, as
    upstream).  `labeled_fns` (SelfSupPartNormalDataset): files left out by base name."""

    def __init__(self, root='./data/shapenetcore_partanno_segmentation_benchmark_v0_normal', npoints=2500, split='train',
                 class_choice=None, normal_channel=False, k_shot=-1, raw=False, rng=None, _labeled_fns=None):
        self.npoints, self.root, self.normal_channel, self.raw, self.k_shot = npoints, root, normal_channel, raw, k_shot
        self._rng = rng
        self.rng = rng if rng is not None else np.random
        self.catfile = os.path.join(root, 'synsetoffset2category.txt')
        self.cat = {}
        with open(self.catfile) as f:
            for line in f:
                ls = line.strip().split()
                if len(ls) >= 2:
                    self.cat[ls[0]] = ls[1]
        self.classes_original = dict(zip(self.cat, range(len(self.cat))))
        if class_choice is not None:
            self.cat = {k: v for k, v in self.cat.items() if k in class_choice}

        def ids(name):
            with open(os.path.join(root, 'train_test_split', name)) as f:
                return set(str(d.split('/')[2]) for d in json.load(f))

        train_ids, val_ids, test_ids = (ids('shuffled_%s_file_list.json' % s) for s in ('train', 'val', 'test'))
        wanted = {'trainval': train_ids | val_ids, 'train': train_ids, 'val': val_ids, 'test': test_ids}
        if _labeled_fns is None:
            wanted['val2'] = test_ids          # a fixed fraction of the test files (:67-70), PartNormalDataset only
        if split not in wanted:
            raise ValueError('Unknown split: %s' % split)
        labeled = set(os.path.basename(x) for x in (_labeled_fns or ()))
        self.datapath = []
        self.meta = {}      # {category: [paths]} -- the trainer builds the self-supervised exclude list from it
        for item in self.cat:
            dir_point = os.path.join(root, self.cat[item])
            fns = [fn for fn in sorted(os.listdir(dir_point)) if fn not in labeled and fn[0:-4] in wanted[split]]
            if split == 'val2':
                fns = _few_shot(fns, round((len(fns) / 2874) * 1870), self._rng)
            if _labeled_fns is None:
                if k_shot > 0 and len(fns) > k_shot:
                    fns = _few_shot(fns, k_shot, self._rng)     # random few-shot subset (:77-79)
            elif k_shot > 0:
                fns = _few_shot(fns, k_shot, self._rng)         # (:208-210: no length guard upstream -- raises when too few)
            self.meta[item] = [os.path.join(dir_point, os.path.splitext(os.path.basename(fn))[0] + '.txt') for fn in fns]
            self.datapath += [(item, fn) for fn in self.meta[item]]
        self.classes = {k: self.classes_original[k] for k in self.cat}
        self.seg_classes = SEG_CLASSES
        self.cache, self.cache_size = {}, 20000

    def _load(self, index):
        if index in self.cache:
            return self.cache[index]
        cat, fn = self.datapath[index]
        cls = np.array([self.classes[cat]]).astype(np.int32)
        data = np.loadtxt(fn).astype(np.float32)
        point_set = data[:, 0:6] if self.normal_channel else data[:, 0:3]
        seg = data[:, -1].astype(np.int32)
        if len(self.cache) < self.cache_size:
            self.cache[index] = (point_set, cls, seg)
        return point_set, cls, seg

    def __getitem__(self, index):
        point_set, cls, seg = self._load(index)
        if self.raw:
            return point_set, cls, seg
        point_set = point_set.copy()
        point_set[:, 0:3] = _normalize_np(point_set[:, 0:3])
        choice = self.rng.choice(len(seg), self.npoints, replace=True)
        return point_set[choice, :], cls, seg[choice]

    def __len__(self):
        return len(self.datapath)


class SelfSupPartNormalDataset(PartNormalDataset):
    """The "dummy" self-supervision set of the trainer (train_partseg_shapenet.py:199-203): the ShapeNet part files that are
    NOT used as labeled data -- `labeled_fns` are left out by base name (ShapeNetDataLoader.py:149-262).  Same items as
    PartNormalDataset: (points [npoints, C], cls [1], seg [npoints])."""

    def __init__(self, root='./data/shapenetcore_partanno_segmentation_benchmark_v0_normal', npoints=2500, split='train',
                 class_choice=None, normal_channel=False, k_shot=-1, labeled_fns=None, raw=False, rng=None):
        if labeled_fns is None:
            raise TypeError("SelfSupPartNormalDataset: labeled_fns (the labeled datasets' file list) is required")
        super().__init__(root, npoints, split, class_choice, normal_channel, k_shot, raw=raw, rng=rng, _labeled_fns=list(labeled_fns))
        self.labeled_files = set(os.path.basename(x) for x in labeled_fns)


class ACDSelfSupDataset(torch.utils.data.Dataset):
    """Self-supervised clouds, `<root>/<subfolder>/<token>.npy`, rows `x y z ... label`
    (ShapeNetDataLoader.py:265-412).  Item: (points [npoints,C], chamfer_points [n,C] = the whole normalised
    cloud, cls [1], seg [npoints])."""

    def __init__(self, root='/srv/data2/mgadelha/ShapeNetACD/', npoints=2500, class_choice=None, normal_channel=False,
                 k_shot=-1, exclude_fns=(), splits=None, use_val=False, prefetch=False, raw=False, rng=None):
        """splits: unused upstream too.  use_val: keep a random 80 % of every sub-folder (:321-323).  prefetch: load, normalise
        and resample every item once, here (:340-366); items are then the stored arrays.  Sub-folders and files are listed
        in SORTED order (upstream takes `os.listdir` order, which is the file system's)."""
        self.npoints, self.root, self.normal_channel, self.raw = npoints, root, normal_channel, raw
        self.k_shot, self.use_val, self.prefetch = k_shot, use_val, prefetch
        self._rng = rng
        self.rng = rng if rng is not None else np.random
        subfolders = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        self.classes_original = dict(zip(subfolders, range(len(subfolders))))
        self.cat = {k: v for k, v in self.classes_original.items() if class_choice is None or k in class_choice}
        # overlap removal compares extension-less tokens: the trainer passes the labeled datasets' '.txt' paths
        # (train_partseg_shapenet.py:194-210) against '.npy' files here (ShapeNetDataLoader.py:305-311)
        exclude = set(os.path.splitext(os.path.basename(x))[0] for x in exclude_fns)
        self.exclude_fns = [os.path.basename(x) for x in exclude_fns]
        self.datapath = []
        self.meta = {}
        for item in self.cat:
            fns = [fn for fn in sorted(os.listdir(os.path.join(root, item)))
                   if fn.endswith('.npy') and os.path.splitext(fn)[0] not in exclude]
            num = len(fns)
            if k_shot > 0 and len(fns) > k_shot:     # (:316-318; upstream's branch stops at an undefined name in its print)
                fns = _few_shot(fns, k_shot, self._rng)
            if use_val:
                fns = _few_shot(fns, math.floor(num * 0.8), self._rng)
            self.meta[item] = [os.path.join(root, item, os.path.splitext(os.path.basename(fn))[0] + '.npy') for fn in fns]
            self.datapath += [(item, fn) for fn in self.meta[item]]
        self.classes = {k: self.classes_original[k] for k in self.cat}
        self.cache = {}
        self.cache_size = len(self.datapath)
        self._items = [self._item(i) for i in range(len(self.datapath))] if (prefetch and not raw) else None

    def _load(self, index):
        if index in self.cache:
            return self.cache[index]
        cat, fn = self.datapath[index]
        cls = np.array([self.classes[cat]]).astype(np.int32)
        data = np.load(fn).astype(np.float32)
        point_set = data[:, 0:6] if self.normal_channel else data[:, 0:3]
        seg = data[:, -1].astype(np.int32)
        self.cache[index] = (point_set, cls, seg)
        return point_set, cls, seg

    def _item(self, index):
        point_set, cls, seg = self._load(index)
        point_set = point_set.copy()
        point_set[:, 0:3] = _normalize_np(point_set[:, 0:3])
        choice = self.rng.choice(len(seg), self.npoints, replace=True)
        return point_set[choice, :], point_set, cls, seg[choice]

    def __getitem__(self, index):
        if self.raw:
            return self._load(index)
        if self._items is not None:
            return self._items[index]
        return self._item(index)

    def __len__(self):
        return len(self.datapath)
