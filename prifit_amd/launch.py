"""One process per GPU: the launcher behind `bench.py --gpus N` (and any other entry point of the package).

The reference scales with one call, `nn.DataParallel(classifier)` (train_partseg_shapenet.py:248-250): one process,
one thread per GPU.  Here every GPU gets its own process (RCCL over xGMI through torch.distributed); this module
starts those processes with the rendezvous environment torch.distributed.run would give them
(RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT).

The PARENT never touches the GPU: it imports neither torch nor the HIP library, so the children are plain
`subprocess` starts (no exec from a GPU-initialised process).  A child that fails takes the job down: the others are
terminated by PID and its exit code is returned.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, world, port, base=None, addr="127.0.0.1"):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR=addr, MASTER_PORT=str(port))
    # dmabuf IPC is the only IPC the host driver supports (RCCL / tensor sharing across processes)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # CPU thread pools of a rank: its share of the cores this job may run on (the rank pins itself to its own cores and
    # sets the exact figure once it knows them: prifit_amd/hostcfg.py); never every core times every rank
    if world > 1:
        share = max(1, len(os.sched_getaffinity(0)) // world)
        env.setdefault("OMP_NUM_THREADS", str(share))
        env.setdefault("MKL_NUM_THREADS", str(share))
    return env


def spawn_ranks(nproc, argv, env=None, port=None, poll=0.05, timeout=None, stdout=None):
    """Start `argv` once per rank (rank r gets RANK=LOCAL_RANK=r), wait for all, return the job's exit code:
    0 if every rank exited 0, else the first non-zero code seen (the remaining ranks are terminated).
    `stdout`: optional list of file objects, one per rank (default: inherit, so rank 0's JSON line reaches the caller)."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = port or free_port()
    procs = []
    for r in range(nproc):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, nproc, port, env),
                                      stdout=None if stdout is None else stdout[r]))
    t0 = time.monotonic()
    rc = 0
    alive = set(range(nproc))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            if alive:
                time.sleep(poll)
    finally:
        for r in sorted(alive):  # exact PIDs of our own children only
            p = procs[r]
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for r in sorted(alive):
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc


def relaunch_self(nproc, script, args):
    """`python script args...` once per rank, with the same interpreter."""
    return spawn_ranks(nproc, [sys.executable, script] + list(args))
