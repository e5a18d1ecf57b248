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
    hands their non-zero exit code back."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
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
