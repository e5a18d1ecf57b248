"""Data-parallel gradient exchange: one flat fp32 bucket, one RCCL all-reduce per optimizer step.

The reference scales with nn.DataParallel (train_partseg_shapenet.py:248-250): one process, one
thread per GPU, parameter broadcast + gradient reduce every iteration, per-replica BatchNorm
statistics.  Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), shapes
sharded by rank, and the 1.76 M parameters' gradients live in ONE contiguous buffer (7 MB) that is
all-reduced once -- the exchange is latency-bound on xGMI, so a single collective is the right
granularity (no per-layer buckets).  BatchNorm statistics stay per-rank (no SyncBN), as in the
reference; `sync_buffers` broadcasts rank 0's running stats ("replica 0 wins" in DataParallel).
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, module, process_group=None):
        self.module = module
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)  # gradients accumulate in place in the bucket
            off += p.numel()
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1

    def zero(self):
        self.flat.zero_()
        off = 0
        for p in self.params:  # re-attach views (a backward may have replaced .grad with a fresh tensor)
            v = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
            off += p.numel()

    def _gather_strays(self):
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
            off += p.numel()

    def allreduce(self):
        """Average gradients over ranks (sum all-reduce of the flat bucket, then scale by 1/world)."""
        self._gather_strays()
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=src, group=self.group)

    def sync_buffers(self, src=0):
        if self.world > 1:
            for b in self.module.buffers():
                dist.broadcast(b.data, src=src, group=self.group)
