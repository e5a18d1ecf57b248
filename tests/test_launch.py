"""CPU: the one-process-per-GPU launcher behind `bench.py --gpus N` (prifit_amd/launch.py), the build's counterpart
of the reference's single `nn.DataParallel(classifier)` call (train_partseg_shapenet.py:248-250)."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prifit_amd import launch  # noqa: E402

ECHO = ("import os,sys; print(' '.join(os.environ[k] for k in "
        "('RANK','LOCAL_RANK','WORLD_SIZE','LOCAL_WORLD_SIZE','MASTER_ADDR','MASTER_PORT','HSA_ENABLE_IPC_MODE_LEGACY')))")


def _spawn(n, code, **kw):
    # the even split of the host's cores is what these tests assert: a *_VISIBLE_DEVICES variable in the children's environment
    # makes hostcfg.rank_cpu_set skip the GPU -> NUMA-node lookup (on a two-socket 8-GPU host ranks 0-3 would otherwise share node
    # 0's cores; that branch has its own test on a faked sysfs tree).  These children never touch a GPU.
    kw.setdefault("env", dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0,1,2,3,4,5,6,7")))
    with tempfile.TemporaryDirectory() as d:
        outs = [open(os.path.join(d, "r%d.txt" % r), "w+") for r in range(n)]
        rc = launch.spawn_ranks(n, [sys.executable, "-c", code], stdout=outs, **kw)
        texts = []
        for f in outs:
            f.seek(0)
            texts.append(f.read())
            f.close()
    return rc, texts


@pytest.mark.timeout(60)
def test_ranks_get_the_rendezvous_environment():
    rc, texts = _spawn(3, ECHO)
    assert rc == 0
    rows = [t.split() for t in texts]
    assert [r[0] for r in rows] == ["0", "1", "2"] and [r[1] for r in rows] == ["0", "1", "2"]
    assert all(r[2] == "3" and r[3] == "3" and r[4] == "127.0.0.1" and r[6] == "0" for r in rows)
    assert len({r[5] for r in rows}) == 1 and int(rows[0][5]) > 0     # one common port


@pytest.mark.timeout(60)
def test_failing_rank_takes_the_job_down():
    # rank 1 fails at once, rank 0 would sleep for a minute: the launcher ends it and reports rank 1's code
    code = "import os,sys,time; r=int(os.environ['RANK']); sys.exit(7) if r==1 else time.sleep(60)"
    rc, _ = _spawn(2, code)
    assert rc == 7


@pytest.mark.timeout(60)
def test_parent_never_imports_torch():
    # importing the launcher and the builder (all the parent of `bench.py --gpus N` imports) must not pull in torch
    code = "import sys; sys.path.insert(0, %r); from prifit_amd import launch, build; assert 'torch' not in sys.modules" % ROOT
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


@pytest.mark.timeout(180)
def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment starts two ranks.  Here (no GPU) each rank stops at
    bench.py's 'needs an MI355X' check -- after reporting its rank through PRIFIT_BENCH_TRACE -- and the launcher
    hands their non-zero exit code back.  (PRIFIT_BENCH_SHARE_GPU=1, the rehearsal switch, lets the launcher start more
    ranks than the GPUs it can count: without it the preflight ends the job first, test_visible_gpu_count_and_preflight.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PRIFIT_BENCH_SHARE_GPU"] = "1"
    with tempfile.TemporaryDirectory() as d:
        env["PRIFIT_BENCH_TRACE"] = os.path.join(d, "trace")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                           env=env, capture_output=True, text=True)
        got = sorted(open(os.path.join(d, f)).read() for f in os.listdir(d))
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gpu-marked rehearsal test")
    assert r.returncode != 0 and "needs an MI355X" in r.stderr
    assert got == ["rank 0 of 2 local 0", "rank 1 of 2 local 1"]


# ---------------------------------------------------------------------------------------------------------------------
# host side of eight ranks (prifit_amd/hostcfg.py): cores per rank, thread caps, the GPU-count preflight
# ---------------------------------------------------------------------------------------------------------------------
def test_rank_cpu_sets_are_disjoint_and_follow_the_gpu_numa_node():
    from prifit_amd import hostcfg
    allowed = list(range(0, 64)) + list(range(128, 192))          # 128 cores the job may use, two sockets' worth
    nodes = {0: list(range(0, 64)), 1: list(range(128, 192))}
    numa_of = lambda r: 0 if r < 4 else 1                          # GPUs 0-3 on node 0, 4-7 on node 1
    sets = [hostcfg.rank_cpu_set(r, 8, allowed=allowed, numa_of=numa_of, node_cpus=lambda n: nodes[n]) for r in range(8)]
    cores = [set(c) for c, _ in sets]
    assert all(len(c) == 16 for c in cores) and len(set().union(*cores)) == 128
    assert all(cores[r] <= set(nodes[0 if r < 4 else 1]) and sets[r][1] == (0 if r < 4 else 1) for r in range(8))
    # no NUMA information: an even split of the allowed cores; fewer cores than ranks: still one core each
    even = [hostcfg.rank_cpu_set(r, 8, allowed=list(range(32)), numa_of=lambda r: None)[0] for r in range(8)]
    assert [len(c) for c in even] == [4] * 8 and len(set().union(*map(set, even))) == 32
    few = [hostcfg.rank_cpu_set(r, 8, allowed=[3, 5], numa_of=lambda r: -1)[0] for r in range(8)]
    assert all(len(c) == 1 and c[0] in (3, 5) for c in few)
    # one node says -1 (unknown): nobody trusts the partial picture
    mixed = [hostcfg.rank_cpu_set(r, 2, allowed=list(range(8)), numa_of=lambda r: (0, -1)[r], node_cpus=lambda n: list(range(8)))
             for r in range(2)]
    assert mixed[0][1] is None and not (set(mixed[0][0]) & set(mixed[1][0]))


@pytest.mark.timeout(60)
def test_rank_env_caps_the_thread_pools_and_ranks_pin_themselves():
    ncores = len(os.sched_getaffinity(0))
    env = launch.rank_env(3, 8, 12345, base={})
    assert env["OMP_NUM_THREADS"] == env["MKL_NUM_THREADS"] == str(max(1, ncores // 8))
    assert "OMP_NUM_THREADS" not in launch.rank_env(0, 1, 12345, base={})
    code = ("import os,sys; sys.path.insert(0, %r); from prifit_amd import hostcfg; h = hostcfg.apply_from_env(set_torch=False); "
            "assert 'torch' not in sys.modules; "
            "print(os.environ['RANK'], ','.join(map(str, sorted(os.sched_getaffinity(0)))), h['threads'], os.environ['OMP_NUM_THREADS'])" % ROOT)
    world = min(4, ncores)
    rc, texts = _spawn(world, code)
    assert rc == 0
    rows = [t.split() for t in texts]
    sets = [set(map(int, r[1].split(","))) for r in rows]
    assert all(len(s) == ncores // world for s in sets) and len(set().union(*sets)) == world * (ncores // world)   # disjoint
    assert all(int(r[2]) == len(s) == int(r[3]) for r, s in zip(rows, sets))


def test_gpu_numa_node_from_a_faked_sysfs(tmp_path, monkeypatch):
    """The GPU -> NUMA node lookup on a faked KFD / DRM tree: nodes 0-1 are CPUs (simd_count 0), nodes 2-3 GPUs with render
    minors 128 / 129 on NUMA nodes 0 / 1; a *_VISIBLE_DEVICES variable (devices reordered) switches the lookup off."""
    from prifit_amd import hostcfg
    kfd, drm = tmp_path / "kfd", tmp_path / "drm"
    for n, (simd, minor) in enumerate([(0, -1), (0, -1), (1216, 128), (1216, 129)]):
        d = kfd / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (64 if simd == 0 else 0, simd, minor))
    for minor, node in ((128, 0), (129, 1)):
        d = drm / ("renderD%d" % minor) / "device"
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    real = hostcfg.kfd_gpu_nodes
    monkeypatch.setattr(hostcfg, "kfd_gpu_nodes", lambda root=str(kfd): real(root))
    assert [g["drm_render_minor"] for g in hostcfg.kfd_gpu_nodes()] == [128, 129]
    assert hostcfg.gpu_numa_node(0, drm_root=str(drm)) == 0 and hostcfg.gpu_numa_node(1, drm_root=str(drm)) == 1
    assert hostcfg.gpu_numa_node(2, drm_root=str(drm)) is None                      # no such GPU
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    assert hostcfg.gpu_numa_node(0, drm_root=str(drm)) is None


def test_visible_gpu_count_and_preflight(monkeypatch):
    from prifit_amd import hostcfg
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert hostcfg.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert hostcfg.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    for v in ("CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    n = hostcfg.visible_gpu_count()
    assert n is None or n >= 0
    # `bench.py --gpus 8` on a box that shows one GPU: one clear line, exit code 2, no rank is started
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = "0"
    with tempfile.TemporaryDirectory() as d:
        env["PRIFIT_BENCH_TRACE"] = os.path.join(d, "trace")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1"],
                           env=env, capture_output=True, text=True)
        assert os.listdir(d) == []
    assert r.returncode == 2 and "--gpus 8 but only 1 GPU(s) are visible" in r.stderr and "Traceback" not in r.stderr


WORLD8 = r'''
import os, sys
sys.path.insert(0, %r)
from prifit_amd import hostcfg
host = hostcfg.apply_from_env(set_torch=False)
import torch, torch.distributed as dist
torch.set_num_threads(host["threads"] or 1)
from prifit_amd.ddp import FlatGradBucket
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(100 + rank)                      # different init per rank on purpose: the broadcast must fix it
net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
bucket = FlatGradBucket(net)
bucket.broadcast_parameters(0)
opt = torch.optim.Adam(net.parameters(), lr=1e-2)
g = torch.Generator().manual_seed(7)
X, Y = torch.randn(4 * world, 6, generator=g), torch.randn(4 * world, 3, generator=g)
xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
for _ in range(3):
    bucket.zero()
    ((net(xs) - ys) ** 2).mean().backward()
    bucket.allreduce()
    opt.step()
bucket.flush()
bucket.sync_buffers(0)
flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.detach().float().reshape(-1) for b in net.buffers()])
got = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(got, flat)
assert all(torch.equal(got[0], t) for t in got), "ranks diverged"
recs = [None] * world
dist.all_gather_object(recs, {"rank": rank, "cores": host["cores"], "threads": host["threads"], "torch_threads": torch.get_num_threads()})
if rank == 0:
    import json
    print(json.dumps({"checksum": float(flat.double().sum()), "ranks": recs}))
dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
def test_world_of_eight_over_gloo_through_the_launcher():
    """Eight ranks started by spawn_ranks (the launcher of `bench.py --gpus 8`), each pinned to its own cores with its
    thread pools capped, exchange gradients through FlatGradBucket for three Adam steps and end bit-identical."""
    import json
    rc, texts = _spawn(8, WORLD8 % ROOT, timeout=240)
    assert rc == 0, texts
    line = json.loads(texts[0].strip().splitlines()[-1])
    ncores = len(os.sched_getaffinity(0))
    per = max(1, ncores // 8)
    assert [r["rank"] for r in line["ranks"]] == list(range(8))
    assert all(len(r["cores"]) == per and r["threads"] == per and r["torch_threads"] == per for r in line["ranks"])
    if ncores >= 8:
        assert len({c for r in line["ranks"] for c in r["cores"]}) == 8 * per                       # disjoint core sets
    assert line["checksum"] == line["checksum"]                                                     # finite
