"""Host side of one-process-per-GPU: which cores a rank runs on and how many CPU threads it may start.

The reference scales with `nn.DataParallel(classifier)` (train_partseg_shapenet.py:248-250): ONE process whose Python thread
launches for every GPU.  Here each GPU has its own process, and the host side of a step (~9 ms of Python per 15 ms of GPU
work) must not be slowed by its neighbours: eight ranks that each start torch with every core's worth of intra-op threads
and float over all sockets are the straggler term of an 8-GPU run.  So every rank, BEFORE its first GPU call,

  * pins itself to its own cores, on the NUMA node of its GPU when sysfs says which that is (an even split of the
    process's affinity mask otherwise),
  * caps its CPU thread pools (torch intra-op, OpenMP, MKL) at that many cores.

Nothing here imports torch at module import or touches the GPU (sysfs reads only), so the launcher parent may use it too.
"""
import glob
import os

_applied = None


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _parse_cpulist(text):
    cpus = []
    for part in (text or "").strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus += list(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def kfd_gpu_nodes(root="/sys/class/kfd/kfd/topology/nodes"):
    """The GPUs of this host in KFD enumeration order (HIP's device order when no *_VISIBLE_DEVICES reorders it):
    a list of dicts {"node", "drm_render_minor"}; [] when the topology is not readable."""
    out = []
    try:
        nodes = sorted((int(os.path.basename(p)), p) for p in glob.glob(os.path.join(root, "*")) if os.path.basename(p).isdigit())
    except OSError:
        return out
    for n, p in nodes:
        props = _read(os.path.join(p, "properties"))
        if props is None:
            continue
        kv = dict(line.split(None, 1) for line in props.splitlines() if len(line.split(None, 1)) == 2)
        if int(kv.get("simd_count", "0") or 0) > 0:
            out.append({"node": n, "drm_render_minor": int(kv.get("drm_render_minor", "-1") or -1)})
    return out


def visible_gpu_count():
    """How many GPUs a rank of this job will see, WITHOUT initialising the runtime: the *_VISIBLE_DEVICES list when one
    is set, else the KFD topology.  None when neither says (then the ranks' own check decides)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    gpus = kfd_gpu_nodes()
    return len(gpus) if gpus else None


def gpu_numa_node(local_rank, drm_root="/sys/class/drm"):
    """NUMA node of the GPU rank `local_rank` drives (-1 / None: unknown).  Only trusted when no *_VISIBLE_DEVICES
    variable reorders the devices."""
    if any(os.environ.get(v) is not None for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")):
        return None
    gpus = kfd_gpu_nodes()
    if local_rank >= len(gpus) or gpus[local_rank]["drm_render_minor"] < 0:
        return None
    t = _read(os.path.join(drm_root, "renderD%d" % gpus[local_rank]["drm_render_minor"], "device", "numa_node"))
    try:
        return int(t)
    except (TypeError, ValueError):
        return None


def rank_cpu_set(local_rank, local_world, allowed=None, numa_of=gpu_numa_node, node_cpus=None):
    """The cores of one rank: (sorted core list, numa node or None).  Ranks whose GPUs sit on the same NUMA node share
    that node's allowed cores evenly, in rank order; with no usable NUMA information every rank gets an even slice of
    the allowed cores.  Always at least one core; the sets of different ranks are disjoint whenever there are at least
    as many allowed cores as ranks."""
    allowed = sorted(os.sched_getaffinity(0) if allowed is None else allowed)
    if node_cpus is None:
        node_cpus = lambda n: _parse_cpulist(_read("/sys/devices/system/node/node%d/cpulist" % n))
    nodes = [numa_of(r) for r in range(local_world)]
    mine = nodes[local_rank]
    if mine is not None and mine >= 0 and all(n is not None and n >= 0 for n in nodes):
        pool = [c for c in node_cpus(mine) if c in set(allowed)]
        peers = [r for r in range(local_world) if nodes[r] == mine]
        if len(pool) >= len(peers):
            per = len(pool) // len(peers)
            i = peers.index(local_rank)
            return pool[i * per:(i + 1) * per], mine
    per = max(1, len(allowed) // max(1, local_world))
    lo = (local_rank * per) % max(1, len(allowed))
    return (allowed[lo:lo + per] or allowed[:1]), None


def apply_rank_affinity(local_rank, local_world, set_torch=True):
    """Pin this process and cap its CPU thread pools (see the module docstring).  Returns what it did:
    {"cores": [...], "threads": n, "numa_node": node or None, "pinned": bool}.  Idempotent per process."""
    global _applied
    if _applied is not None and _applied["local_rank"] == local_rank and _applied["local_world"] == local_world:
        return _applied
    cores, node = rank_cpu_set(local_rank, local_world)
    pinned = False
    if local_world > 1:
        try:
            os.sched_setaffinity(0, cores)
            pinned = True
        except OSError:
            pass
    threads = max(1, len(cores)) if local_world > 1 else None
    if threads is not None:
        # a value the user (or launch.rank_env) set explicitly wins; the cap never RAISES a pool above it
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            try:
                preset = int(os.environ[var])
            except (KeyError, ValueError):
                preset = None
            if preset is not None and preset > 0:
                threads = min(threads, preset) if var == "OMP_NUM_THREADS" else threads
            else:
                os.environ[var] = str(threads)
        if set_torch:
            import torch
            torch.set_num_threads(threads)
    _applied = {"local_rank": local_rank, "local_world": local_world, "cores": list(cores), "threads": threads,
                "numa_node": node, "pinned": pinned}
    return _applied


def apply_from_env(set_torch=True):
    """apply_rank_affinity for the rank the rendezvous environment describes.  The LOCAL size of the job decides how the
    host's cores are divided: LOCAL_RANK / LOCAL_WORLD_SIZE (torch.distributed.run and launch.rank_env set both).  Without
    LOCAL_WORLD_SIZE the global RANK / WORLD_SIZE are used only when the job is known to sit on one node (no LOCAL_RANK, or
    LOCAL_RANK == RANK); a multi-node rank without its local size is left alone rather than given 1 / WORLD_SIZE of this
    host's cores.  A single process is left alone.  NOTE: pinning reaches threads created AFTER the call -- call it before
    importing torch (bench.py does; Trainer calls it late and then only caps the pools)."""
    if "LOCAL_WORLD_SIZE" in os.environ:
        world = int(os.environ["LOCAL_WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", "0"))
    else:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if local != rank:        # several nodes and no local size: do not guess
            world, local = 1, 0
    return apply_rank_affinity(local, world, set_torch=set_torch)
