"""GPU box (one MI355X): bench.py's three launch paths run the same step.
  plain            python bench.py                       (no process group)
  1-rank RCCL      RANK/WORLD_SIZE set by the launcher   (pack, all-reduce, broadcast, barrier over RCCL)
  2-rank rehearsal `bench.py --gpus 2` with PRIFIT_DIST_BACKEND=gloo PRIFIT_BENCH_SHARE_GPU=1: the launcher really
                   starts two ranks (both on the one GPU, gradients exchanged over gloo), the line says n_gpus 2 and is
                   marked as a rehearsal.  The real thing (RCCL, one GPU per rank) needs a multi-GPU node."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
ARGS = ["--steps", "2", "--warmup", "2", "--workload", "c2", "--no-cpu-baseline"]


def _line(cmd, env):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 1, r.stdout[-2000:]
    return json.loads(rows[0])


@pytest.mark.timeout(1500)
def test_bench_launch_paths_agree(hiplib):
    sys.path.insert(0, ROOT)
    from prifit_amd import launch
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    plain = _line([sys.executable, bench, "--gpus", "1"] + ARGS, base)
    assert plain["n_gpus"] == 1 and plain["config"]["global_batch"] == 24
    one = _line([sys.executable, bench, "--gpus", "1"] + ARGS, launch.rank_env(0, 1, launch.free_port(), base))
    assert one["n_gpus"] == 1
    # the N > 1 line proves itself: world size and backend from the communicator, who ran where
    d1 = one["distributed"]
    assert d1["world_size"] == 1 and d1["backend"] == "nccl" and d1["distinct_gpus"] == 1
    r0 = d1["ranks"][0]
    assert r0["rank"] == 0 and r0["device_index"] == 0 and (r0["pci_bus_id"] or r0["uuid"]) and r0["ms_per_step"] > 0
    assert one["allreduce_ms_per_step"] >= 0 and "distributed" not in plain
    # same seeds, same step: the loss after 4 steps agrees (split-K float atomics: reproducible to rounding only)
    assert abs(one["config"]["loss"] - plain["config"]["loss"]) < 1e-3 * abs(plain["config"]["loss"])
    env = dict(base, PRIFIT_DIST_BACKEND="gloo", PRIFIT_BENCH_SHARE_GPU="1")
    with tempfile.TemporaryDirectory() as d:
        env["PRIFIT_BENCH_TRACE"] = os.path.join(d, "t")
        two = _line([sys.executable, bench, "--gpus", "2"] + ARGS, env)
        assert sorted(open(os.path.join(d, f)).read() for f in os.listdir(d)) == ["rank 0 of 2 local 0", "rank 1 of 2 local 1"]
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 48 and two["config"]["parallelism"] == "dp2"
    assert "rehearsal" in two and two["value"] > 0
    d2 = two["distributed"]
    assert d2["world_size"] == 2 and d2["backend"] == "gloo" and [r["rank"] for r in d2["ranks"]] == [0, 1]
    assert d2["distinct_gpus"] == 1            # both ranks on the one GPU: accepted only because the line is a labelled rehearsal
    assert d2["ms_per_step_min"] <= d2["ms_per_step_max"] and len(d2["speculation_fallbacks_per_rank"]) == 2
    assert two["allreduce_ms_per_step"] > 0    # the exchange really ran (event-bracketed, per step)


@pytest.mark.timeout(600)
def test_native_rccl_allreduce_one_rank(hiplib):
    """prifit_allreduce_flat (the C-ABI RCCL export) with a one-rank communicator on the one GPU of this box:
    unique id -> communicator -> in-place sum all-reduce on the current stream = identity, through NativeComm and
    through FlatGradBucket(native=True).  Runs in a child process (its own process group)."""
    code = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from prifit_amd import launch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()))
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
from prifit_amd.rccl import NativeComm
from prifit_amd.ddp import FlatGradBucket
c = NativeComm.from_process_group()
x = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
y = x.clone()
c.allreduce_(y)
torch.cuda.synchronize()
assert torch.equal(x, y)
net = torch.nn.Linear(8, 4).cuda()
b = FlatGradBucket(net, native=True)
assert b.native is not None
net(torch.randn(3, 8, device="cuda")).sum().backward()
g = net.weight.grad.clone()
b.allreduce()
torch.cuda.synchronize()
assert torch.equal(net.weight.grad, g)
c.destroy(); b.native.destroy()
print("native rccl ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "native rccl ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.timeout(900)
def test_labelled_experiment_line_says_what_it_is(hiplib):
    """`bench.py --ms-split MODE` (the 16-bit-planes experiment of the mean-shift forward): the line carries the label in
    `dtype` and `experiment`, the split kernel is the dominant family, no `extra`; without the flag `dtype` is plain f32."""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    base["PRIFIT_BENCH_EVENTS"] = "all"     # every kernel family in `kernels` (by default only the dominant one and the grouping launches)
    bench = os.path.join(ROOT, "bench.py")
    args = ["--steps", "2", "--warmup", "2", "--workload", "c3", "--no-cpu-baseline", "--no-extra"]
    exp = _line([sys.executable, bench] + args + ["--ms-split", "fp16x3"], base)
    assert exp["dtype"].startswith("f32 (mean-shift forward products fp16x3 emulated") and "experiment" in exp
    assert "extra" not in exp and any(k.startswith("ms_split_fwd[fp16x3]") for k in exp["kernels"])
    assert not any(k.startswith("ms_fused_fwd") for k in exp["kernels"])
    plain = _line([sys.executable, bench] + args, base)
    assert plain["dtype"] == "f32" and "experiment" not in plain
    assert any(k.startswith("ms_fused_fwd") for k in plain["kernels"]) and not any(k.startswith("ms_split") for k in plain["kernels"])
    assert abs(exp["config"]["loss"] - plain["config"]["loss"]) < 1e-3 * abs(plain["config"]["loss"])
