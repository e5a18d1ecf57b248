#!/usr/bin/env python
"""Headline benchmark of the PRIFIT hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--cloud cube|blobs|surface]
                    [--embedding clustered|untrained|clustered25|retry40] [--no-extra]

One "step" = one training iteration of the hot path over one batch of B=24 synthetic 2048-point
clouds per GPU: zero_grad, forward, loss, backward, gradient all-reduce (N>1, RCCL), Adam step.
  c2: PointNet++-MSG part-seg, segmentation loss only          (BASELINE.json configs[1])
  c3: c2's network + mean-shift (10 it, <=25 clusters) + ellipsoid fit + convex loss (configs[2],
      the configuration the metric is quoted on) -- the self-supervised step of
      train_partseg_shapenet.py:436-451.
  c5: DGCNN backbone (k=20) + the same fit path                (configs[4])
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

The headline (`value`) is measured on synthetic clouds with a seeded network whose embedding carries 8 part prototypes per
shape (`--embedding clustered`, synth.part_embedding_offset: ~8 clusters per shape, the reference's regime is up to 25,
README.md:62).  The seeded untrained network ALONE collapses to ONE cluster per shape: membership is identically 1, the loss
does not depend on the embedding and the gradient into the whole backbone is exactly zero -- that condition is
kept as `extra.untrained_embedding`, never `value`.  After the timed region the line asserts that the first backbone layer
received a gradient (config.backbone_grad_ratio = |grad W| / |W| >= 1e-6; the one-cluster condition gives exactly 0).  At N = 1 the c3 line also carries `extra`: the
same step on surface clouds, at K = 25, with a retry in every step, the untrained condition, and short c2 / c5 runs --
shapes/s, clusters per shape, speculation fallbacks and the fit-path kernel rows of each.  `--cloud` / `--embedding` make one
of them THE measured condition (then named in config).

`--gpus N` with N > 1 and no RANK in the environment: this process only LAUNCHES -- it starts N copies of
itself, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (prifit_amd/launch.py), before anything
has touched the GPU, waits for them and exits with their code.  Under `python -m torch.distributed.run` (RANK
already set) it is one of the ranks.  The reference's counterpart is `nn.DataParallel(classifier)`
(train_partseg_shapenet.py:248-250).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# NOTE: torch is imported inside the functions: the launcher parent must stay off the GPU, and the oracle/
# directory is put on the path by the cpu_baseline leg only (it is the CPU baseline being timed there).

B_PER_GPU = 24
NPTS = 2048
NUM_PARTS = 50
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2516.6  # same table: dense bf16 / fp16 MFMA = 16 x the fp32 rate (only the labelled split-products experiment)
HBM_PEAK_GBS = 8000.0           # HBM3E spec (6.29 TB/s measured by a float4 copy)
# launches of the ball-query + grouping stage: sa_group_linear = ball query + grouping + first MLP layer of a
# set-abstraction level in one launch (default); the others run with PRIFIT_SA_FUSED=0 / PRIFIT_SA_LINEARITY=0
GROUPING_FAMILIES = ("sa_group_linear", "ball_query", "group_gather", "gather_linear")
WORKLOADS = {
    "c2": "configs[1]: PointNet++-MSG part-seg, B=24x2048 per GPU, seg loss only, fwd+bwd+Adam",
    "c3": "configs[2]: PointNet++-MSG + mean-shift(10 it, <=25 clusters) + ellipsoid fit + convex loss, "
          "B=24x2048 per GPU, fwd+bwd+Adam",
    "c5": "configs[4]: DGCNN (k=20) + mean-shift(10 it, <=25 clusters) + ellipsoid fit + convex loss, "
          "B=24x2048 per GPU, fwd+bwd+Adam",
}


# --embedding conditions beyond "clustered": name -> (equal-size spatial parts per shape as (nx, ny), prototype noise, mean-shift quantile)
# (25 parts, measured round 5 over the 35 Adam steps of the measurement: noise 0.03 / q 0.02 -- the bandwidth is wide enough for
# far parts to pull modes together, K drifts 14..25 and 3 of 30 steps retry; noise 0.01 / q 0.02 -- the network's own spatially
# smooth output stretches a part into two modes, 22 of 30 steps retry; in between, isotropic noise dominates)
EMBEDDING_PARTS = {"clustered25": ((5, 5), 0.02, 0.03), "retry40": ((5, 8), 0.005, 0.01)}
DEFAULT_CLOUD = {"c2": "cube", "c3": "blobs", "c5": "blobs"}   # uniform cube; blobs where clusters must exist
# The measured condition of the fit workloads is the CLUSTERED embedding: with the seeded untrained network
# alone mean-shift finds one cluster per shape, the membership weights are identically 1 and the gradient that reaches the
# backbone is exactly zero -- a step that multiplies zeros.  8 part prototypes per shape added to the network's own embedding
# (synth.part_embedding_offset) give K ~ 8 and real gradients through every layer; parity of exactly this step at B = 24:
# tests/test_gpu_bench_step_parity.py[clustered].  `untrained` stays as an `extra` entry.
DEFAULT_EMBEDDING = {"c2": "untrained", "c3": "clustered", "c5": "clustered"}
# |grad W| / |W| of the FIRST backbone layer after the last timed step, asserted outside the timed region.  Measured (round 6):
# c3 clustered 4e-5 (first steps) .. 5e-4 (after 25 Adam steps), surface clouds 6e-5, c5 7e-5, c2 8e-3, K = 25 4e-6; the one-cluster
# condition gives EXACTLY 0 on the HIP path and <= 1e-8 (rounding noise) in the oracle -- the bar sits between the two.
MIN_BACKBONE_GRAD_RATIO = 1e-6
METRIC = {"c2": "shapes/sec (fwd+bwd) B=24x2048 pts, PointNet++-MSG seg loss only",
          "c3": "shapes/sec (fwd+bwd) B=24x2048 pts, PointNet++-MSG+ellipsoid fit",
          "c5": "shapes/sec (fwd+bwd) B=24x2048 pts, DGCNN+ellipsoid fit"}


def make_inputs(workload, rank, device, cloud=None):
    """cloud: cube | blobs | surface (default per workload).  c3 / c5 also get the 5000 chamfer targets the model input is
    a fixed 2048-subset of (train_partseg_shapenet.py:441) and `parts`, a spatial 8-part labelling of the input points
    (the blobs' generating labels / Voronoi cells) for --embedding clustered."""
    import numpy as np
    import torch
    from prifit_amd import synth
    seed = 1000 * rank  # seed 0 on rank 0
    cloud = cloud or DEFAULT_CLOUD[workload]
    if workload == "c2":
        xyz = torch.from_numpy(synth.cloud(cloud, B_PER_GPU, NPTS, seed)).transpose(1, 2).contiguous()
        d = {"xyz": xyz, "target": torch.from_numpy(synth.labels(B_PER_GPU, NPTS, NUM_PARTS, seed))}
    else:
        sel = np.random.default_rng(seed + 1).choice(5000, NPTS, replace=False)
        if cloud == "blobs":
            cham_np, lab = synth.blobs_with_labels(B_PER_GPU, 5000, seed)     # == synth.cloud("blobs", ...) + its labels
            parts = lab[:, sel]
        else:
            cham_np = synth.cloud(cloud, B_PER_GPU, 5000, seed)
            parts = synth.part_labels(cham_np[:, sel], 8, seed)
        cham = torch.from_numpy(cham_np)
        d = {"xyz": cham[:, sel].transpose(1, 2).contiguous(), "chamfer": cham.transpose(1, 2).contiguous(),
             "parts": torch.from_numpy(parts)}
    d["cls"] = torch.zeros(B_PER_GPU, 1, 16)
    d["s1"] = torch.from_numpy(synth.fps_start(B_PER_GPU, NPTS, seed))
    d["s2"] = torch.from_numpy(synth.fps_start(B_PER_GPU, 512, seed + 100))
    return {k: v.to(device) for k, v in d.items()}


def build_model(device, workload="c3"):
    import torch
    from prifit_amd import synth
    torch.manual_seed(0)
    if workload == "c5":
        from prifit_amd.src import dgcnn as D
        return D.get_model(NUM_PARTS, k=20).to(device).train(), None
    from prifit_amd.models import pointnet2_part_seg_msg as M
    net = M.get_model(NUM_PARTS)
    synth.xavier_like_trainer(net)  # train_partseg_shapenet.py:240-247
    return net.to(device).train(), M


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, Bs=4, passes=5):
    """The oracle (CPU restatement of the reference, kind "port") on a bounded sample of the same
    workload: B=4 shapes, 1 warm-up + 5 timed forward+backward passes (median, min and max reported), host
    cores of this box.
    (`--cpu-baseline-shapes 24` times the full batch once: B = 24 measured 1.05x the shapes/s of B = 4)"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import prifit_oracle as orc
    from prifit_amd import synth
    cores = min(os.cpu_count() or 1, 32)  # torch-CPU oversubscribes badly beyond ~32 threads on these small ops
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    xyz = torch.from_numpy(synth.cloud("cube" if workload == "c2" else "blobs", Bs, NPTS, 0)).transpose(1, 2).contiguous()
    cls = torch.zeros(Bs, 1, 16)
    target = torch.from_numpy(synth.labels(Bs, NPTS, NUM_PARTS, 0))
    s = (torch.from_numpy(synth.fps_start(Bs, NPTS, 0)), torch.from_numpy(synth.fps_start(Bs, 512, 100)))
    fit = dict(quantile=0.05, iterations=10, max_num_clusters=25)
    if workload != "c2":
        cham_np, lab = synth.blobs_with_labels(Bs, 5000, 0)
        cham = torch.from_numpy(cham_np)
        sel = np.random.default_rng(1).choice(5000, NPTS, replace=False)
        xyz = cham[:, sel].transpose(1, 2).contiguous()
        cham_t = cham.transpose(1, 2).contiguous()
        # the condition the GPU line measures (DEFAULT_EMBEDDING: 8 part prototypes per shape added to the embedding)
        if DEFAULT_EMBEDDING[workload] == "clustered":
            fit["embedding_offset"] = torch.from_numpy(synth.part_embedding_offset(lab[:, sel], 128, 0))
    if workload == "c5":
        net = orc.OracleDGCNGn(emb_size=128, num_channels=3, nn_nb=20).train()

        def one():
            net.zero_grad()
            emb, _ = net(xyz)                                  # [B,N,128] (src/dgcnn.py:225-267)
            total = orc.convex_loss(xyz, cham_t, emb.transpose(1, 2), **fit)[0]
            total.mean().backward()
    else:
        net = orc.OracleMSGPartSeg(NUM_PARTS)
        synth.xavier_like_trainer(net)
        net.train()
        extra = {}
        if workload == "c3":
            extra = dict(chamfer_points=cham_t, include_convex_loss=True, quantile=0.05, msc_iterations=10,
                         max_num_clusters=25)
            if "embedding_offset" in fit:
                extra["fit_inputs"] = dict(embedding_offset=fit["embedding_offset"])

        def one():
            net.zero_grad()
            out = net(xyz, cls, fps_start=s, **extra)
            loss = orc.seg_loss(out[0].reshape(-1, NUM_PARTS), target.view(-1)) if workload == "c2" else out[3].mean()
            loss.backward()

    one()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    # a baseline, not a target: the figure moves with the box's other tenants (1.1 .. 2.1 shapes/s were seen for the same
    # sample on one afternoon), so the line carries the spread of its passes, not a point
    return {"value": Bs / t, "unit": "shapes/s", "cores": cores, "kind": "port", "cpu": _cpu_model(), "B": Bs,
            "passes": passes, "seconds_per_pass": t, "value_min": Bs / max(ts), "value_max": Bs / min(ts),
            "seconds_per_pass_all": [round(x, 4) for x in ts],
            "sample": "oracle/prifit_oracle.py (torch-CPU restatement of the reference, %d threads on %s), B=%d x %d "
                      "pts, %s step fwd+bwd, median of %d timed passes after 1 warm-up (%.2f s per pass, min %.2f, max %.2f)"
                      % (cores, _cpu_model(), Bs, NPTS, workload, passes, t, min(ts), max(ts))}


def _traffic_per_launch(dom):
    """HBM bytes per launch of the dominant family from the committed PMC passes (rocprofv3 cannot run inside this
    process): profiles/r0X_pmc_traffic.json, FETCH_SIZE x2 + WRITE_SIZE, see the file's "source"."""
    # (the PMC tool names kernels, the spans name call sites)
    alias = {"ms_fused_fwd": "ms_fused_kernel<0, true, false, 4, 2, false", "ms_first_fwd": "ms_fused_kernel<0, true, false, 4, 2, true", "ms_fused_bwd": "ms_fused_kernel<1, true, true",
             "gemm_dual_nn": "gemm_dual_sk_kernel"}
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                fams_pmc = json.load(fh)["families"]
            rec = fams_pmc.get(dom)
            if rec is None and dom in alias:   # kernel names carry further template arguments: match the prefix
                rec = next((v for k, v in fams_pmc.items() if k.startswith(alias[dom])), None)
            rec = rec or {}
            if rec.get("hbm_bytes_per_launch") is not None:
                return rec["hbm_bytes_per_launch"], name
        except Exception:
            continue
    return None, None


def device_identity(index):
    """What tells two GPUs apart: PCI address and uuid of visible device `index` (torch's device properties; the HIP
    runtime's hipDeviceGetPCIBusId as a fallback)."""
    import torch
    ident = {"device_index": int(index), "pci_bus_id": None, "uuid": None, "name": None}
    try:
        pr = torch.cuda.get_device_properties(index)
        ident["name"] = pr.name
        if hasattr(pr, "uuid"):
            ident["uuid"] = str(pr.uuid)
        if hasattr(pr, "pci_bus_id"):
            ident["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    except Exception:   # noqa: BLE001
        pass
    if ident["pci_bus_id"] is None:
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
                ident["pci_bus_id"] = buf.value.decode()
        except Exception:   # noqa: BLE001
            pass
    return ident


def distributed_report(mine, rehearsal=False, group=None):
    """COLLECTIVE (every rank calls it): what makes an N > 1 line prove itself.  `mine` = this rank's record (rank,
    local_rank, host, the device identity, its own ms_per_step, speculation_fallbacks, allreduce_ms_per_step).  Returns on
    every rank {"world_size" (from the communicator, not from the environment), "backend", "ranks": [...all records, by
    rank...], "distinct_gpus", "ms_per_step_min" / "_max" over ranks}; raises unless the ranks sit on `world_size`
    distinct GPUs -- except in a labelled rehearsal (ranks sharing a GPU, PRIFIT_BENCH_SHARE_GPU=1)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    recs = [None] * world
    dist.all_gather_object(recs, mine, group=group)
    recs.sort(key=lambda r: r["rank"])
    if [r["rank"] for r in recs] != list(range(world)):
        raise RuntimeError("distributed_report: ranks %s of a world of %d" % ([r["rank"] for r in recs], world))
    gpus = {(r.get("host"), r.get("pci_bus_id") or r.get("uuid") or ("index", r.get("device_index"))) for r in recs}
    rep = {"world_size": world, "backend": dist.get_backend(group), "ranks": recs, "distinct_gpus": len(gpus),
           "ms_per_step_min": min(r["ms_per_step"] for r in recs), "ms_per_step_max": max(r["ms_per_step"] for r in recs),
           "allreduce_ms_per_step_max": max((r.get("allreduce_ms_per_step") or 0.0) for r in recs),
           "speculation_fallbacks_per_rank": [r.get("speculation_fallbacks") for r in recs]}
    if len(gpus) != world and not rehearsal:
        raise RuntimeError("%d ranks on %d distinct GPUs: %s" % (world, len(gpus), sorted(map(str, gpus))))
    return rep


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    trace = os.environ.get("PRIFIT_BENCH_TRACE")     # tests: each rank leaves a line saying who it is
    if trace:   # written BEFORE torch is imported (seconds on a cold box): a rank the launcher ends early has left it
        with open("%s.%d" % (trace, rank), "w") as f:
            f.write("rank %d of %d local %d" % (rank, world, local))
    # host side of one process per GPU (prifit_amd/hostcfg.py): this rank's own cores (on its GPU's NUMA node when sysfs
    # says which) and CPU thread pools capped at that many -- BEFORE torch starts its pools and before any GPU call
    from prifit_amd import hostcfg
    host = hostcfg.apply_from_env(set_torch=False)
    import torch
    import torch.distributed as dist
    if host["threads"]:
        torch.set_num_threads(host["threads"])

    if "RANK" in os.environ and args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the environment wins" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # PRIFIT_BENCH_SHARE_GPU=1 (rehearsal of the N>1 code path on a one-GPU box, with PRIFIT_DIST_BACKEND=gloo):
    # ranks share the visible devices round-robin.  Never used for a reported number.
    ndev = torch.cuda.device_count()
    share = os.environ.get("PRIFIT_BENCH_SHARE_GPU", "0") == "1"
    if local >= ndev and not share:
        raise SystemExit("bench.py: rank %d has LOCAL_RANK %d but only %d GPU(s) are visible (--gpus %d needs %d)"
                         % (rank, local, ndev, args.gpus, world))
    local_dev = local % ndev
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    # under torch.distributed.run / the launcher (RANK set) the RCCL path is taken even with one rank, so that a
    # 1-GPU box rehearses exactly the code the N>1 runs execute (pack, all-reduce, broadcast, barrier)
    use_dist = world > 1 or "RANK" in os.environ
    backend = os.environ.get("PRIFIT_DIST_BACKEND", "nccl")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from prifit_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_library()   # file-locked + atomic rename: safe when every rank gets here at once

    if use_dist:
        world = dist.get_world_size()     # the communicator's answer is the one the line reports (n_gpus), not WORLD_SIZE
        rank = dist.get_rank()
    ctx = {"world": world, "rank": rank, "local": local, "device": device, "use_dist": use_dist, "backend": backend,
           "share": share}
    cloud = args.cloud or DEFAULT_CLOUD[args.workload]
    head = measure(args, ctx, cloud, args.embedding, args.steps, args.warmup, full=True, split=args.ms_split)

    report = None
    if use_dist:
        import socket
        mine = dict(device_identity(local_dev), rank=rank, local_rank=local, host=socket.gethostname(),
                    cpu_cores=host["cores"], cpu_threads=host["threads"], numa_node=host["numa_node"], cpu_pinned=host["pinned"],
                    ms_per_step=1e3 * head["elapsed_local"] / head["steps"], speculation_fallbacks=head["fallbacks"],
                    allreduce_ms_per_step=head["allreduce_ms"] / head["steps"])
        report = distributed_report(mine, rehearsal=share)
    if rank == 0:
        line = headline(args, ctx, head, cloud)
        # the measured step must DO the backbone's backward: a condition whose loss does not depend on
        # the embedding (one cluster per shape) multiplies zeros -- refuse to report it as `value` unless asked for by name
        if head["grad_ratio"] < MIN_BACKBONE_GRAD_RATIO and not (args.embedding == "untrained" and args.workload != "c2"):
            raise SystemExit("bench.py: |grad %s| / |W| = %.3e < %.0e after the last timed step: the loss did not reach the "
                             "backbone (clusters per shape: %s)" % (head["grad_param"], head["grad_ratio"],
                                                                    MIN_BACKBONE_GRAD_RATIO, line["config"]["clusters_per_shape"]))
        if report is not None:
            line["distributed"] = report
            line["n_gpus"] = report["world_size"]
            line["allreduce_ms_per_step"] = report["allreduce_ms_per_step_max"]
        if args.ms_split != "0":
            line["dtype"] = SPLIT_DTYPE % args.ms_split
            line["experiment"] = SPLIT_NOTE
        if backend != "nccl" and use_dist:
            line["rehearsal"] = "backend=%s%s: NOT a reportable number" % (backend, ", ranks share GPUs" if share else "")
    # the same step under other conditions, beside the headline (single GPU, c3, default condition only)
    if world == 1 and args.workload == "c3" and not args.no_extra and args.default_condition and args.ms_split == "0":
        # (the headline is measured and stays: an exception in a side measurement is recorded in its entry, not raised)
        extra = {}
        short = max(10, min(args.steps, 30))
        for name, cl, emb, wl in (("untrained_embedding", "blobs", "untrained", None),
                                  ("surface_cloud_clustered_embedding", "surface", "clustered", None),
                                  ("clusters_at_the_cap_25_parts", "blobs", "clustered25", None),
                                  ("retry_every_step_40_parts_q0.01", "blobs", "retry40", None),
                                  ("c2_seg_loss_only", DEFAULT_CLOUD["c2"], DEFAULT_EMBEDDING["c2"], "c2"),
                                  ("c5_dgcnn_fit", DEFAULT_CLOUD["c5"], DEFAULT_EMBEDDING["c5"], "c5")):
            try:
                r = measure(args, ctx, cl, emb, min(short, 20) if wl else short, 5, full=False, workload=wl)
                extra[name] = condition_summary(r, cl, emb, wl or args.workload)
            except Exception as e:   # noqa: BLE001
                extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        # LABELLED EXPERIMENT beside the headline (never the headline): the headline's own condition with the mean-shift
        # forward's two products on the 16-bit matrix pipe, error-compensated (csrc/meanshift_split.hip)
        exp = {"note": SPLIT_NOTE}
        for mode in ("fp16x3", "bf16x6"):
            try:
                r = measure(args, ctx, cloud, args.embedding, max(10, min(args.steps, 30)), 5, full=False, split=mode)
                c = condition_summary(r, cloud, args.embedding, args.workload)
                rows = family_rows(r["fams_all"], r["fams_all_steps"])
                exp[mode] = {"dtype": SPLIT_DTYPE % mode, "value": c["value"], "unit": "shapes/s", "ms_per_step": c["ms_per_step"],
                             "vs_fp32_headline": c["value"] / line["value"], "loss": c["loss"],
                             "clusters_per_shape": c["clusters_per_shape"], "speculation_fallbacks": c["speculation_fallbacks"],
                             "kernel": {k: v for k, v in rows.items() if k.startswith("ms_split_fwd")}}
            except Exception as e:   # noqa: BLE001
                exp[mode] = {"error": "%s: %s" % (type(e).__name__, e)}
        extra["split_mean_shift_products_experiment"] = exp
        line["extra"] = extra
        # the training-like conditions at the top level of the line (the driver's parsed record keeps top-level keys)
        line["extra_summary"] = {k: {"value": v.get("value"), "unit": "shapes/s", "ms_per_step": v.get("ms_per_step"),
                                     "workload": v.get("workload"), "launches_per_step": v.get("launches_per_step"),
                                     "backbone_grad_ratio": v.get("backbone_grad_ratio"),
                                     "clusters_per_shape_mean": (v.get("clusters_per_shape") or {}).get("mean"),
                                     "speculation_fallbacks": v.get("speculation_fallbacks"), "error": v.get("error")}
                                 for k, v in extra.items() if k != "split_mean_shift_products_experiment"}
    if rank == 0:
        if world == 1 and os.environ.get("PRIFIT_BENCH_COUNT_LAUNCHES", "1") != "0":
            try:   # after every timed measurement of this process
                line["launches_per_step"] = count_launches(head["_step"])
            except Exception as e:   # noqa: BLE001
                line["launches_per_step"] = None
                line["launches_per_step_error"] = "%s: %s" % (type(e).__name__, e)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_baseline_shapes,
                                                    passes=5 if args.cpu_baseline_shapes <= 8 else 2)
            except Exception as e:   # noqa: BLE001 (the oracle is test infrastructure: its failure must not cost the measured line)
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


SPLIT_DTYPE = "f32 (mean-shift forward products %s emulated: LABELLED EXPERIMENT, not the fp32 product path)"
SPLIT_NOTE = ("the S = Z X^T and O = K X products of the ten mean-shift updates run on the 16-bit matrix pipe with every fp32 "
              "operand cut into 2 (bf16x3, fp16x3) or 3 (bf16x6) 16-bit planes and the significant plane products accumulated "
              "in fp32 (csrc/meanshift_split.hip); exponent, clamp, row sums, the normalisation and the whole backward stay "
              "fp32.  Passes test_convex_loss_end_to_end, test_selfsup_step_matches_reference_golden and the mean-shift golden "
              "at UNCHANGED tolerances (bf16x6, fp16x3); error of ten updates against fp64 next to the fp32 kernel's: "
              "profiles/r03_split_products.json.  Never the headline: `value` of this line is the fp32 path.")
FIT_FAMILIES = ("chord_sym", "chord_sym_mask", "kth_smallest", "ms_first_fwd", "ms_fused_fwd", "ms_split_fwd", "ms_rows_bwd", "ms_fused_bwd", "gemm_dual_nn", "nms", "membership",
                "ellipsoid_fit", "sdf", "sample_nn", "sample_nn_bwd")


def measure(args, ctx, cloud, embedding, steps, warmup, full, split="0", workload=None):
    """Build the network and the inputs of one condition, run `warmup` untimed + `steps` timed training steps bracketed by
    barrier + synchronize, and return the raw measurements.  full: the headline's extras (enqueue time, all ranks' MAX).
    split: the LABELLED EXPERIMENT of csrc/meanshift_split.hip for this measurement ("0" = the fp32 product path).
    workload: another workload than args.workload (the short c2 / c5 side runs of the c3 line)."""
    from prifit_amd import fit_ops
    fit_ops.MS_SPLIT = split
    if workload is not None and workload != args.workload:
        args = argparse.Namespace(**dict(vars(args), workload=workload))
    try:
        return _measure(args, ctx, cloud, embedding, steps, warmup, full)
    finally:
        fit_ops.MS_SPLIT = "0"
        gc.enable()   # (_measure switches the cyclic collector off around its timed region)


def _measure(args, ctx, cloud, embedding, steps, warmup, full):
    import torch
    import torch.distributed as dist
    from prifit_amd import profiler, synth
    from prifit_amd.ddp import FlatGradBucket
    from prifit_amd.train_step import SpeculativeRunner

    world, rank, device, use_dist = ctx["world"], ctx["rank"], ctx["device"], ctx["use_dist"]
    net, M = build_model(device, args.workload)
    bucket = FlatGradBucket(net)
    bucket.broadcast_parameters(0)
    flat_adam = os.environ.get("PRIFIT_FLAT_ADAM", "1") != "0"
    if flat_adam:
        from prifit_amd.optim import FlatAdam     # one launch per step (csrc/optim.hip); 0: torch's fused Adam (A/B)
        opt = FlatAdam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    else:
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, fused=True)
    data = make_inputs(args.workload, rank, device, cloud)
    crit = M.get_loss() if M is not None else None
    runner = SpeculativeRunner(net)
    fit_kw = dict(chamfer_points=data.get("chamfer"), include_convex_loss=True, quantile=0.05, msc_iterations=10,
                  max_num_clusters=25)
    if embedding != "untrained" and args.workload == "c2":
        raise SystemExit("--embedding %s: a workload with the fit path (c3 / c5)" % embedding)
    if embedding == "clustered":
        # what training does to the embedding, as an explicit input: 8 part prototypes per shape (synth.part_embedding_offset)
        off = torch.from_numpy(synth.part_embedding_offset(data["parts"].cpu().numpy(), 128, 1000 * rank)).to(device)
        fit_kw["fit_inputs"] = dict(embedding_offset=off)
    elif embedding in EMBEDDING_PARTS:
        # the fit path where it is loaded: K at the `max_num_clusters` cap (25 spatial parts per shape, README.md:62 regime), or
        # 40 tight parts at quantile 0.01 -- more modes than the cap, so EVERY step takes guard_mean_shift's
        # quantile-doubling retry (src/ellipsoid_utils.py:19-27: full recompute per doubling)
        (nx, ny), noise, q = EMBEDDING_PARTS[embedding]
        parts = synth.equal_part_labels(data["xyz"].transpose(1, 2).cpu().numpy(), nx, ny)
        # (scale 100, not the 30 of "clustered": at 30 the network's own output is 0.15 of a prototype's length, as much as the
        # prototype noise, and as Adam moves it over the measurement's 35 steps a part splits into two modes every few steps:
        # 26 modes = a retry; measured round 5: 10 of 30 steps)
        off = torch.from_numpy(synth.part_embedding_offset(parts, 128, 1000 * rank, K=nx * ny, noise=noise, scale=100.0)).to(device)
        fit_kw["fit_inputs"] = dict(embedding_offset=off)
        fit_kw["quantile"] = q

    # Farthest-point sampling of the NEXT batch on a side stream while this step runs (ops.sample_ahead: the samples depend
    # on the coordinates alone, and the search is 640 serial rounds on one workgroup per shape: 24 of 256 CUs busy for
    # 0.28 ms at the head of every step).  Every step launches the sampling of one batch -- the synthetic batch is the same
    # every step, its samples are recomputed every step all the same -- and consumes the one launched a step earlier: the
    # work inside the timed region is unchanged, only its place in the step.  PRIFIT_SAMPLE_AHEAD = 2 (default): launched
    # right behind the backbone forward (`net.after_backbone`), where it runs beside the matrix-bound mean-shift kernels
    # (c3 16.27 -> 16.08 ms, same box, alternating runs; grouping launches unaffected: 0.51 both ways); 1: between
    # forward and backward (16.10); 0: in line on the step's own stream.  Round 2 had it in line: launched at the top of the
    # step it ran beside the HBM-bound grouping launches, which lost 7 %.
    ahead_mode = os.environ.get("PRIFIT_SAMPLE_AHEAD", "2")
    ahead_on = args.workload != "c5" and ahead_mode in ("1", "2")
    starts = (data.get("s1"), data.get("s2"))
    sampled = {"cur": starts, "next": None}
    last = {}

    def sample_next(force=False):
        if ahead_on and (force or ahead_mode == "1"):
            sampled["next"] = net.sample_ahead(data["xyz"], starts)   # the next step's batch

    if ahead_on and ahead_mode == "2":
        net.after_backbone = lambda: sample_next(True)

    def selfsup_fwd_bwd():
        if args.workload == "c5":
            out = net(data["xyz"], None, **fit_kw)
        else:
            out = net(data["xyz"], data["cls"], fps_start=sampled["cur"], **fit_kw)
        loss = out[3].mean()
        last["count"] = out[6].count      # clusters per shape (device tensor; read after the timed region)
        sample_next()
        loss.backward()
        return loss

    def step():
        if ahead_on:
            sampled["cur"] = sampled["next"] if sampled["next"] is not None else starts
        bucket.zero()
        if args.workload == "c2":
            seg = net(data["xyz"], data["cls"], fps_start=sampled["cur"])[0]
            sample_next()
            loss = crit(seg.reshape(-1, NUM_PARTS), data["target"].view(-1), None)
            loss.backward()
        elif os.environ.get("PRIFIT_SPECULATE", "1") != "0":
            # the cluster-count verdict of guard_mean_shift is read after the whole step is enqueued; a step whose
            # verdict asks for the quantile-doubling retry is discarded and re-run synchronously (counted below)
            loss = runner.run(selfsup_fwd_bwd, bucket.zero)
        else:
            loss = selfsup_fwd_bwd()
        if use_dist and ar_events is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            bucket.allreduce()
            b.record()
            ar_events.append((a, b))
        else:
            bucket.allreduce()
        if flat_adam:
            opt.step(grads=bucket.grads())
        else:
            opt.step()
        return loss

    ar_events = None   # a list inside the timed region: the (pack + all-reduce + scale) span of every step, by HIP events

    # Warm-up steps also calibrate the profiler: every kernel family is bracketed with HIP events once, then
    # only the dominant family and the ball-query/grouping launches keep their events in the timed region
    # (an event pair per launch costs host time, ~7 ms/step when all ~250 wrapped launches are bracketed).
    # (calibration on the LAST warm-up step only: the first launch of every kernel includes the lazy load of its
    # code object, which an event bracket would charge to that family)
    # (a span also brackets the host code between its two event records: a cyclic-GC pause of the launch thread inside one
    # -- 60 ms were seen in a 2-launch family -- would crown the wrong family.  So the collector is off from here on, and
    # the calibration is the per-family MINIMUM of two bracketed steps (the first launch of a kernel also loads its code).)
    ncal = 2   # (so at least two untimed steps run whatever --warmup says; the line reports the number that ran)
    warmup = max(warmup, ncal)
    if args.workload != "c2" and os.environ.get("PRIFIT_SPECULATE", "1") != "0":
        # The first warm-up step also goes through the fall-back of the speculative runner once (re-run with the synchronous
        # clustering), whatever its verdict: the first fall-back of a process grows the caching allocator by ~4 GB (0.2 s of
        # hipMalloc, tools/step_outliers.py) -- one-time initialisation like the lazy code-object loads, which would
        # otherwise be charged to whichever timed step happens to fall back first (one in 200 on some runs, none on most).
        warmup = max(warmup, ncal + 1)
        runner.force_next = True
    for _ in range(warmup - ncal):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    cal = None
    for _ in range(ncal):
        profiler.reset()
        profiler.enable("*")
        step()
        one = profiler.collect()
        profiler.disable()
        cal = one if cal is None else {k: (v if k not in cal or v[1] < cal[k][1] else cal[k]) for k, v in one.items()}
    dominant = max(cal.items(), key=lambda kv: kv[1][1])[0] if cal else None
    profiler.reset()
    graph_note = None
    if args.graph:
        if args.workload == "c5":
            raise SystemExit("--graph: the PointNet++ backbone (c2 / c3)")
        from prifit_amd.train_step import graph_backbone
        graph_backbone(net, data["xyz"], data["cls"], starts)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        graph_note = ("backbone forward + backward replayed as two HIP graphs (train_step.graph_backbone); the grouping "
                      "launches are inside them: roofline_grouping from the last eager warm-up step")
    if os.environ.get("PRIFIT_BENCH_EVENTS", "1") == "all":   # diagnosis: every family bracketed (slows the step down)
        profiler.enable("*")
    elif os.environ.get("PRIFIT_BENCH_EVENTS", "1") != "0":  # 0: diagnosis only (no roofline objects in the line)
        profiler.enable(*[n for n in (dominant,) + GROUPING_FAMILIES if n])
    # no cyclic-GC pauses inside the timed region (a generation-2 pass over the autograd graphs stalls the launch
    # thread for tens of ms; nothing on the step relies on the cycle collector, see MeanShiftFn.forward)
    gc.collect()
    gc.disable()
    fallbacks0 = runner.fallbacks
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ar_events = [] if use_dist else None
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0     # this rank's own time (the reported one is the MAX behind the barrier)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) if ar_events else 0.0
    ar_events = None
    bucket.flush()     # the deferred has-gradient check of the last exchange (ddp.FlatGradBucket)
    gc.enable()
    profiler.disable()
    # OUTSIDE the timed region: did the last timed step send a gradient into the backbone?  (first weight matrix of the
    # network = the first set-abstraction / edge-convolution layer; after the exchange, so the averaged gradient at N > 1)
    wname, w0 = next((n, p) for n, p in net.named_parameters() if p.dim() >= 2)
    grad_ratio = float(w0.grad.norm() / w0.detach().norm()) if w0.grad is not None else 0.0
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    fams = profiler.collect()
    if graph_note:   # the launches inside the graphs carry no events: their rows come from the calibration step (one eager step)
        for k in GROUPING_FAMILIES:
            if k in cal and k not in fams:
                n, ms, work = cal[k]
                fams[k] = (n * steps, ms * steps, work * steps)
    res = {"elapsed": el.item(), "elapsed_local": elapsed_local, "allreduce_ms": allreduce_ms, "steps": steps, "warmup": warmup, "fams": fams, "loss": float(loss.item()), "graph": graph_note,
           "fallbacks": runner.fallbacks - fallbacks0, "ahead_on": ahead_on, "grad_ratio": grad_ratio, "grad_param": wname,
           "_step": step,
           "clusters": last["count"].tolist() if "count" in last else None}
    if full:
        # host time to ENQUEUE one step, measured outside the timed region from an empty queue (inside it the launch
        # thread runs ahead until the queue is full and then advances at the GPU's pace, which says nothing)
        # (c3 / c5: the host then BLOCKS until the GPU reaches the clustering verdict of this very step, ~8 ms in -- that wait is
        # not enqueue work and is taken out: fit_ops.spec_wait_s; reported beside it)
        from prifit_amd import fit_ops as _fo
        t_host, t_wait = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            w0 = _fo.spec_wait_s
            h0 = time.perf_counter()
            step()
            dt = time.perf_counter() - h0
            t_wait.append(_fo.spec_wait_s - w0)
            t_host.append(dt - t_wait[-1])
        torch.cuda.synchronize()
        res["t_host"] = sorted(t_host)[2]
        res["t_host_wait"] = sorted(t_wait)[2]
    else:
        # every kernel family bracketed on two more (untimed) steps: the fit-path rows of this condition
        profiler.reset()
        profiler.enable("*")
        for _ in range(2):
            step()
        res["fams_all"] = profiler.collect()
        res["fams_all_steps"] = 2
        profiler.disable()
        profiler.reset()
    if os.environ.get("PRIFIT_BENCH_CENSUS"):   # diagnosis, outside the timed region: who launches the small torch kernels
        launch_census(step, os.environ["PRIFIT_BENCH_CENSUS"])
    return res


def count_launches(step):
    """Device kernels + memsets / copies of ONE step (torch.profiler's device activity: the same records rocprofv3 sees).
    Called after every timed measurement of the process is over."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    dev_type = getattr(torch.autograd, "DeviceType", None)
    n = 0
    for ev in prof.events():
        if dev_type is not None and getattr(ev, "device_type", None) == dev_type.CUDA:
            n += 1
    return n


def launch_census(step, path):
    """One step under torch.profiler with Python stacks: every device kernel with its count and time, and for the torch
    glue (fill / copy / cat / elementwise) the source line of this repo that issued it.  Written to `path` as text."""
    import collections
    import torch
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
    here = os.path.dirname(os.path.abspath(__file__))
    by_site = collections.defaultdict(lambda: [0, 0.0, set()])
    kernels = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.kernels:
            continue
        dev_us = sum(k.duration for k in ev.kernels)
        for k in ev.kernels:
            kernels[k.name[:90]][0] += 1
            kernels[k.name[:90]][1] += k.duration
        if not ev.name.startswith("aten::"):
            continue
        site = "(no Python stack: autograd thread)" if not ev.stack else "?"
        for fr in ev.stack or ():
            if ("prifit_amd/" in fr or "bench.py" in fr) and "/torch/" not in fr:
                site = fr[fr.find("prifit_amd/"):] if "prifit_amd/" in fr else fr[fr.find("bench.py"):]
                break
        rec = by_site[(site, ev.name)]
        rec[0] += len(ev.kernels)
        rec[1] += dev_us
    with open(path, "w") as f:
        f.write("== device kernels of one step\n")
        for name, (n, us) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
            f.write("%-92s %5d %10.1f us\n" % (name, n, us))
        f.write("\n== torch operators that launched kernels, by the repo line that called them\n")
        for (site, op), (n, us, _) in sorted(by_site.items(), key=lambda kv: -kv[1][1]):
            f.write("%-70s %-28s %4d %9.1f us\n" % (site[:70], op, n, us))


def family_rows(fams, steps):
    """Per kernel family: launches / ms per step, achieved rate against the roofline that bounds it."""
    detail = {}
    for name, (n, ms, work) in sorted(fams.items(), key=lambda kv: -kv[1][1]):
        per = {"launches_per_step": n / steps, "ms_per_step": ms / steps, "avg_us": 1e3 * ms / max(n, 1)}
        base = name.split("[")[0]
        if base == "ms_split_fwd":   # experiment: `work` is the fp32-equivalent product; the pipe executes 3 or 6 plane products
            terms = 6 if "x6" in name else 3
            per.update(bound="mfma", achieved=terms * work / (ms * 1e-3) / 1e12, peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                       flops_per_step=terms * work / steps, flops_per_launch=terms * work / max(n, 1),
                       fp32_equivalent_tflops=work / (ms * 1e-3) / 1e12, plane_products=terms)
        elif base in ("sample_nn", "knn3_topk"):    # pairwise searches on the fp32 vector ALU (its peak = the fp32 MFMA peak on gfx950)
            per.update(bound="valu", achieved=work / (ms * 1e-3) / 1e12, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                       flops_per_step=work / steps, flops_per_launch=work / max(n, 1))
        elif (base.startswith("gemm") and not base.startswith("gemm_stream")) or base.startswith(("ms_fused", "ms_first", "chord_sym")):
            per.update(bound="mfma", achieved=work / (ms * 1e-3) / 1e12, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                       flops_per_step=work / steps, flops_per_launch=work / max(n, 1))
        else:
            per.update(bound="hbm", achieved=work / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                       bytes_per_step=work / steps, bytes_per_launch=work / max(n, 1))
        per["frac"] = per["achieved"] / per["peak"]
        detail[name] = per
    return detail


def grouping_roofline(detail):
    grp = [detail[k] for k in detail if k.split("[")[0] in GROUPING_FAMILIES]
    if not grp:
        return None
    ms = sum(g["ms_per_step"] for g in grp)
    gb = sum(g["achieved"] * g["ms_per_step"] * 1e-3 for g in grp)
    # compulsory traffic of ball query + MATERIALISED grouping of the MSG backbone per shape
    # (the work the reference does) -- THE headline fraction of the >= 0.5-of-HBM target
    sa1 = 12 * (2048 + 512) + sum(4 * 512 * k + 4 * 512 * k * 6 for k in (32, 64, 128))
    sa2 = 12 * (512 + 128) + 4 * 512 * 320 + sum(4 * 128 * k + 4 * 128 * k * 323 for k in (64, 128))
    survey_gb = B_PER_GPU * (sa1 + sa2) / 1e9
    return {"bound": "hbm", "achieved": survey_gb / (ms * 1e-3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": survey_gb / (ms * 1e-3) / HBM_PEAK_GBS, "ms_per_step": ms,
            "gbytes_per_step": survey_gb,
            "note": "SURVEY.md 8(d) formula: bytes of the reference's ball query + materialised grouping "
                    "(858 MB per B=24 batch) / time of the launches that now do that job (both set-abstraction levels; SA2's "
                    "first layer by linearity needs two small pre-GEMMs, U and Vc, 2 x 31 us, that replace a 49 GFLOP GEMM "
                    "and are counted with the MLP, not here: 0.40 with them)",
            "own_bytes": {"gbytes_per_step": gb, "achieved": gb / (ms * 1e-3),
                          "frac": gb / (ms * 1e-3) / HBM_PEAK_GBS,
                          "note": "bytes the launches really move (sa_group_linear = ball query + grouping + "
                                  "first MLP layer in one launch: clouds, index lists, C1-wide first-layer "
                                  "rows, U / Vc)"}}


def condition_summary(r, cloud, embedding, workload="c3"):
    """One `extra` entry: the step under another condition / workload."""
    ms = 1e3 * r["elapsed"] / r["steps"]
    ks = r["clusters"] or []
    hist = {}
    for k in ks:
        hist[str(int(k))] = hist.get(str(int(k)), 0) + 1
    rows = family_rows(r["fams_all"], r["fams_all_steps"])
    grouping = grouping_roofline(rows)
    return {"workload": workload, "cloud": cloud, "embedding": embedding if workload != "c2" else None,
            "quantile": EMBEDDING_PARTS.get(embedding, (None, None, 0.05))[2] if workload != "c2" else None,
            "backbone_grad_ratio": r.get("grad_ratio"), "launches_per_step": r.get("launches_per_step"),
            "value": B_PER_GPU * r["steps"] / r["elapsed"], "unit": "shapes/s",
            "ms_per_step": ms, "steps": r["steps"], "warmup": r["warmup"], "loss": r["loss"],
            "clusters_per_shape": {"mean": (sum(ks) / len(ks)) if ks else None, "min": min(ks) if ks else None,
                                   "max": max(ks) if ks else None, "histogram": hist},
            "speculation_fallbacks": r["fallbacks"],
            "grouping_ms_per_step": grouping["ms_per_step"] if grouping else None,
            "grouping_frac_survey_formula": grouping["frac"] if grouping else None,
            "fit_path_kernels": {k: {"launches_per_step": v["launches_per_step"], "ms_per_step": v["ms_per_step"],
                                     "avg_us": v["avg_us"]}
                                 for k, v in rows.items() if k.split("[")[0] in FIT_FAMILIES},
            "note": "kernel rows from 2 extra steps with every family bracketed by HIP events (not inside the timed region)"}


def headline(args, ctx, r, cloud):
    world = ctx["world"]
    steps = r["steps"]
    ms_per_step = 1e3 * r["elapsed"] / steps
    value = world * B_PER_GPU * steps / r["elapsed"]
    detail = family_rows(r["fams"], steps)
    roof = None
    if detail:
        dom = next(iter(detail))   # dominant kernel family by accumulated event time
        d = detail[dom]
        traffic, traffic_src = _traffic_per_launch(dom)
        roof = {"kernel": dom, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"],
                "frac": d["frac"], "traffic": traffic, "traffic_source": traffic_src, "avg_us": d["avg_us"],
                "launches_per_step": d["launches_per_step"]}
        # the numerator, so that the line can be re-derived: achieved = work_per_launch / avg_us
        for k in ("flops_per_step", "flops_per_launch", "bytes_per_step", "bytes_per_launch"):
            if k in d:
                roof[k] = d[k]
    # guard against stale work models: no family may print more than its roofline (a model that still charges work a
    # fusion removed, or full products where a triangle is computed, shows up here)
    over = {k: round(v["frac"], 3) for k, v in detail.items() if v["frac"] > 1.0}
    if over:
        print("bench.py: work model above its roofline (stale model?): %s" % over, file=sys.stderr)
    grouping = grouping_roofline(detail)
    if grouping:
        pmc, src = _traffic_per_launch("sa_group_linear")
        n = sum(v["launches_per_step"] for k, v in detail.items() if k.split("[")[0] == "sa_group_linear")
        grouping["own_bytes"]["pmc_gbytes_per_step"] = (pmc * n / 1e9) if (pmc and n) else None
        grouping["own_bytes"]["pmc_source"] = src
    ks = r["clusters"]
    line = {
        "metric": METRIC[args.workload],
        "value": value, "unit": "shapes/s", "n_gpus": world, "steps": steps, "warmup": r["warmup"],
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload],
                   "global_batch": world * B_PER_GPU, "points": NPTS, "parallelism": "dp%d" % world,
                   "cloud": cloud, "embedding": args.embedding if args.workload != "c2" else None,
                   "clusters_per_shape": (sum(ks) / len(ks)) if ks else None,
                   "backbone_grad_ratio": r["grad_ratio"], "backbone_grad_param": r["grad_param"],
                   "loss": r["loss"],
                   "launch": r.get("graph") or "eager",
                   "fps": ("side stream, one batch ahead: every step launches one batch's sampling (behind its backbone forward) "
                           "and consumes the previous launch; PRIFIT_SAMPLE_AHEAD=0 runs it in line" if r["ahead_on"] else "in line")},
        "roofline": roof, "roofline_grouping": grouping, "kernels": detail, "roofline_model_violations": over,
        "speculation_fallbacks": r["fallbacks"],
        "host_enqueue_ms_per_step": 1e3 * r["t_host"],
        "host_verdict_wait_ms_per_step": 1e3 * r.get("t_host_wait", 0.0),
    }
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 10 warm-up steps bring the clocks and the caching allocator to steady state (with 3 the first timed
    # steps still run ~10 % slow); 200 timed steps = ~5 s of GPU work (a sampled GPU-busy monitor sees it beside the ~15 s
    # CPU-baseline leg; 50 steps measure the same rate to 0.3 %)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("PRIFIT_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-shapes", type=int, default=4, help="shapes of the CPU-baseline sample (4: ~15 s)")
    # the measured condition (defaults = the headline: BASELINE.json's synthetic clouds, seeded untrained network)
    ap.add_argument("--cloud", default=None, choices=("cube", "blobs", "surface"))
    ap.add_argument("--embedding", default=None, choices=("untrained", "clustered") + tuple(sorted(EMBEDDING_PARTS)),
                    help="default: clustered for c3 / c5 (~8 clusters per shape: real gradients into the backbone), n/a for c2")
    ap.add_argument("--no-extra", action="store_true", help="skip the training-like conditions reported under `extra`")
    ap.add_argument("--graph", action="store_true", help="replay the backbone forward + backward as HIP graphs (static shapes)")
    ap.add_argument("--ms-split", default="0", choices=("0", "bf16x3", "bf16x6", "fp16x3"),
                    help="LABELLED EXPERIMENT: mean-shift forward products on the 16-bit matrix pipe, error-compensated; the line's "
                         "dtype says so and it is never the reported fp32 number")
    args = ap.parse_args()
    args.default_condition = args.embedding is None and args.cloud is None
    if args.embedding is None:
        args.embedding = DEFAULT_EMBEDDING[args.workload]

    if args.gpus > 1 and "RANK" not in os.environ:
        # launcher: nothing in this branch imports torch or loads the HIP library
        from prifit_amd import build, hostcfg, launch
        # preflight: one clear line instead of N tracebacks when the box has fewer GPUs than ranks were asked for
        # (sysfs / *_VISIBLE_DEVICES only: the parent stays off the GPU)
        have = hostcfg.visible_gpu_count()
        if have is not None and have < args.gpus and os.environ.get("PRIFIT_BENCH_SHARE_GPU", "0") != "1":
            print("bench.py: --gpus %d but only %d GPU(s) are visible on this host" % (args.gpus, have), file=sys.stderr)
            sys.exit(2)
        build.build_library()            # once, before the ranks start (hipcc only; a no-op when up to date)
        sys.exit(launch.relaunch_self(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
