"""CPU, world_size 2 over gloo: the flat-bucket gradient exchange of prifit_amd.ddp (the N>1 path of
bench.py) averages gradients exactly like one big batch, keeps parameters in lock-step, and
`sync_buffers` makes rank 0's BatchNorm statistics win (DataParallel semantics)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prifit_amd.ddp import FlatGradBucket

    torch.manual_seed(100 + rank)  # different init per rank on purpose: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    bucket = FlatGradBucket(net)
    bucket.broadcast_parameters(0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randn(8, 3, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]  # shard by rank, equal sizes
    for _ in range(3):
        bucket.zero()
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        bucket.allreduce()
        opt.step()
    bucket.sync_buffers(0)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.detach().float().reshape(-1) for b in net.buffers()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        bucket.pack()
        out.put([t.numpy().copy() for t in gathered] + [bucket.flat.numpy().copy()])  # by value, not shared memory
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=100)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    r0, r1, grad = (torch.from_numpy(a) for a in res)
    assert torch.equal(r0, r1), "ranks diverged"
    assert torch.isfinite(grad).all() and grad.abs().sum() > 0


def test_single_process_bucket_matches_plain_autograd():
    sys.path.insert(0, ROOT)
    from prifit_amd.ddp import FlatGradBucket
    torch.manual_seed(0)
    net = torch.nn.Linear(5, 4)
    ref = torch.nn.Linear(5, 4)
    ref.load_state_dict(net.state_dict())
    bucket = FlatGradBucket(net)
    x = torch.randn(3, 5)
    for _ in range(2):
        bucket.zero()
        ref.zero_grad()
        net(x).sum().backward()
        ref(x).sum().backward()
        bucket.allreduce()  # single rank: gradients stay where autograd put them
        assert torch.equal(net.weight.grad, ref.weight.grad)
        bucket.pack()
        assert torch.equal(net.weight.grad, ref.weight.grad) and net.weight.grad.data_ptr() == bucket.flat.data_ptr()
