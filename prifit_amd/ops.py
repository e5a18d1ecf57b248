"""Thin functional wrappers over the C ABI: index / grouping ops of the PointNet++ stack.

Every function takes and returns CUDA(HIP) tensors in channels-last layout and launches on the
current PyTorch stream.  Reference sites are cited per function (paths relative to upstream).
"""
import ctypes

import numpy as np
import torch

from . import _lib, profiler
from . import arena as zero_pool
from ._lib import call, cf, cur_stream, ptr, require_cuda


class SampledAhead:
    """Farthest-point samples of one set-abstraction level computed ahead of the step that uses them (`sample_ahead`):
    pass it where a module takes `fps_start`.  `farthest_point_sample` then only makes the consuming stream wait for the
    event and hands the stored (indices, coordinates) out -- the same numbers the in-line launch would produce."""
    __slots__ = ("idx", "new_xyz", "event", "npoint", "keep", "consumed", "stream")

    def __init__(self, idx, new_xyz, event, keep=(), stream=None):
        self.idx, self.new_xyz, self.event, self.npoint = idx, new_xyz, event, idx.shape[1]
        self.keep = keep   # the side stream's inputs: not back to the allocator before the consumer has waited for the event
        self.consumed = False   # set by farthest_point_sample once the consuming stream waits for the event
        # the stream whose allocator pool idx / new_xyz came from (the current stream when sample_ahead ran): a finaliser
        # can run anywhere -- inside `with torch.cuda.stream(side)`, on another thread -- so "current stream" at that
        # moment is not the pool's stream
        self.stream = stream

    def __del__(self):
        # dropped WITHOUT having been consumed: the side stream may still be writing idx / new_xyz, whose memory goes back to
        # the consuming stream's allocator pool right now -- make that stream wait for the chain (a device-side wait: no
        # host sync in a finaliser).  Once consumed, the consumer's wait_event already orders every later reuse.
        if self.consumed:
            return
        try:
            if not self.event.query():
                (self.stream or torch.cuda.current_stream(self.idx.device)).wait_event(self.event)
        except Exception:   # interpreter shutdown
            pass


_ahead_streams = {}


def sample_ahead(xyz, npoints, starts=None):
    """The farthest-point-sampling CHAIN of a batch (level l samples npoints[l] of level l - 1's samples; it depends on
    the coordinates alone) on a side stream, for the NEXT step's batch while the current step runs: the search is a
    serial loop of `npoint` rounds on one workgroup per shape (24 of 256 CUs busy for 0.32 ms of a 24 ms step), so on
    the step's own stream the rest of the chip waits for it.  xyz [B,N,3] channels-last; returns one SampledAhead per
    level.  The side stream first waits for everything already enqueued on the current stream (the batch's producers)."""
    require_cuda(xyz)
    dev = xyz.device
    side = _ahead_streams.get(dev)
    if side is None:
        side = _ahead_streams[dev] = torch.cuda.Stream(dev)
    xyz = cf(xyz)
    B = xyz.shape[0]
    starts = list(starts) if starts is not None else [None] * len(npoints)
    n_in = [xyz.shape[1]] + [int(n) for n in npoints[:-1]]
    starts = [torch.randint(0, n, (B,), dtype=torch.long, device=dev) if s is None
              else s.to(device=dev, dtype=torch.int64).contiguous() for s, n in zip(starts, n_in)]
    # outputs are allocated on the CONSUMING stream's pool: they are freed there, after the wait for the event
    outs = [(torch.empty(B, int(n), dtype=torch.int64, device=dev), torch.empty(B, int(n), 3, dtype=torch.float32, device=dev))
            for n in npoints]
    consumer = torch.cuda.current_stream(dev)
    side.wait_stream(consumer)
    with torch.cuda.stream(side):
        cur = xyz
        for (idx, nx), st, n in zip(outs, starts, npoints):
            call("prifit_fps", ptr(cur), B, cur.shape[1], int(n), ptr(st), ptr(idx), ptr(nx), cur_stream())
            cur = nx
        ev = torch.cuda.Event()
        ev.record(side)
    return [SampledAhead(idx, nx, ev, keep=(xyz, starts), stream=consumer) for idx, nx in outs]


def farthest_point_sample(xyz, npoint, start_idx=None, return_xyz=False):
    """models/pointnet_util.py:63-84.  xyz [B,N,3] -> int64 [B,npoint] (bit-exact for a given
    start_idx; when None a random start is drawn like the reference's torch.randint at :75).
    `start_idx` may be a SampledAhead (the samples of this level, launched earlier by `sample_ahead`)."""
    require_cuda(xyz)
    if isinstance(start_idx, SampledAhead):
        if start_idx.npoint != npoint or start_idx.idx.shape[0] != xyz.shape[0]:
            raise ValueError("SampledAhead holds %s samples, the module asks for [%d, %d]"
                             % (tuple(start_idx.idx.shape), xyz.shape[0], npoint))
        torch.cuda.current_stream(xyz.device).wait_event(start_idx.event)
        start_idx.consumed = True
        return (start_idx.idx, start_idx.new_xyz) if return_xyz else start_idx.idx
    xyz = cf(xyz)
    B, N, _ = xyz.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)
    start_idx = start_idx.to(device=xyz.device, dtype=torch.int64).contiguous()
    out = torch.empty(B, npoint, dtype=torch.int64, device=xyz.device)
    new_xyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device) if return_xyz else None
    call("prifit_fps", ptr(xyz), B, N, npoint, ptr(start_idx), ptr(out), ptr(new_xyz), cur_stream())
    return (out, new_xyz) if return_xyz else out


def ball_query_multi(radius_list, nsample_list, xyz, new_xyz, idx64=False):
    """models/pointnet_util.py:87-107 for several radii in one pass.  Returns a list of
    [B,S,nsample] index tensors (int32 internally, int64 for the reference-surface API)."""
    require_cuda(xyz, new_xyz)
    xyz, new_xyz = cf(xyz), cf(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    outs = []
    for lo in range(0, len(radius_list), 4):
        rs = radius_list[lo:lo + 4]
        ks = nsample_list[lo:lo + 4]
        R = len(rs)
        o = [torch.empty(B, S, k, dtype=torch.int64 if idx64 else torch.int32, device=xyz.device) for k in ks]
        # the reference compares fp32 distances with the python double radius**2, i.e. fp32(radius**2)
        r2 = (ctypes.c_float * R)(*[float(np.float32(r ** 2)) for r in rs])
        ns = (ctypes.c_int * R)(*[int(k) for k in ks])
        op = (ctypes.c_void_p * R)(*[t.data_ptr() for t in o])
        # algorithmic bytes: clouds read once, indices written once
        with profiler.span("ball_query", B * (12.0 * (N + S) + sum(4.0 * S * k for k in ks))):
            call("prifit_ball_query", ptr(xyz), ptr(new_xyz), B, N, S, R, r2, ns, op, int(idx64), cur_stream())
        outs += o
    return outs


def three_nn(xyz1, xyz2, want_dist=False):
    """models/pointnet_util.py:291-297: 3 nearest of xyz2 for every xyz1 point + normalised
    inverse-distance weights.  Returns (idx int32 [B,N,3], weight [B,N,3][, dist [B,N,3]])."""
    require_cuda(xyz1, xyz2)
    xyz1, xyz2 = cf(xyz1), cf(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = torch.empty(B, N, 3, dtype=torch.int32, device=xyz1.device)
    w = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device)
    d = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device) if want_dist else None
    call("prifit_three_nn", ptr(xyz1), ptr(xyz2), B, N, S, ptr(idx), ptr(d), ptr(w), cur_stream())
    return (idx, w, d) if want_dist else (idx, w)


def square_distance(src, dst):
    """models/pointnet_util.py:19-40 (expanded form, bitwise)."""
    require_cuda(src, dst)
    src, dst = cf(src), cf(dst)
    B, S, _ = src.shape
    N = dst.shape[1]
    out = torch.empty(B, S, N, dtype=torch.float32, device=src.device)
    call("prifit_square_distance", ptr(src), ptr(dst), B, S, N, ptr(out), cur_stream())
    return out


def group_gather(feat, xyz, new_xyz, idx, order=0, ld_out=None):
    """models/pointnet_util.py:243-249 / :127-133.  Returns [B*S*K, ld_out] rows
    [feat, rel_xyz, 0..] (order 0, MSG) or [rel_xyz, feat, 0..] (order 1, SSG)."""
    require_cuda(xyz, new_xyz, idx)
    xyz, new_xyz = cf(xyz), cf(new_xyz)
    feat = None if feat is None else cf(feat)
    idx = idx.contiguous()
    assert idx.dtype == torch.int32
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    C = 0 if feat is None else feat.shape[-1]
    if ld_out is None:
        ld_out = (C + 3 + 3) // 4 * 4
    out = torch.empty(B * S * K, ld_out, dtype=torch.float32, device=xyz.device)
    # algorithmic bytes: feature table read once + grouped rows [S*K, C+3] written once
    with profiler.span("group_gather", B * (4.0 * N * C + 4.0 * S * K * (C + 3))):
        call("prifit_group_gather", ptr(feat), ptr(xyz), ptr(new_xyz), ptr(idx), B, N, S, K, C, order, ld_out,
             ptr(out), cur_stream())
    return out


def group_scatter_add(gout, col0, idx, B, N, C, dfeat=None):
    """Backward of the feature columns of group_gather: returns dfeat [B,N,C]."""
    _, S, K = idx.shape
    if dfeat is None:
        dfeat = zero_pool.zeros(B, N, C, device=gout.device)
    call("prifit_group_scatter_add", ptr(gout), gout.stride(0), col0, ptr(idx), B, N, S, K, C, ptr(dfeat),
         cur_stream())
    return dfeat


def three_interpolate(points2, idx, weight, out=None, col0=0):
    """models/pointnet_util.py:298.  points2 [B,S,C] -> out[(b,n), col0:col0+C]."""
    B, S, C = points2.shape
    N = idx.shape[1]
    if out is None:
        out = torch.empty(B * N, C, dtype=torch.float32, device=points2.device)
    call("prifit_three_interpolate", ptr(points2), ptr(idx), ptr(weight), B, N, S, C, out.stride(0), col0,
         ptr(out), cur_stream())
    return out


def three_interpolate_bwd(gout, col0, idx, weight, B, S, C):
    N = idx.shape[1]
    dp2 = zero_pool.zeros(B, S, C, device=gout.device)
    call("prifit_three_interpolate_bwd", ptr(gout), gout.stride(0), col0, ptr(idx), ptr(weight), B, N, S, C,
         ptr(dp2), cur_stream())
    return dp2
