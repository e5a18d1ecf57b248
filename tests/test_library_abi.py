"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares."""
import ctypes
import os
import subprocess

from prifit_amd import _lib


def test_header_symbols_exported(hiplib):
    names = _lib.declared_symbols()
    assert len(names) >= 9
    for n in names:
        assert hasattr(hiplib, n), n


def test_version_and_arch(hiplib):
    arch = ctypes.c_char_p()
    v = hiplib.prifit_version(ctypes.byref(arch))
    assert v >= 100 and arch.value == b"gfx950"


def test_code_object_is_gfx950():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH], capture_output=True,
                         text=True)
    # the fat binary embeds an amdgcn-amd-amdhsa--gfx950 code object
    with open(_lib.LIB_PATH, "rb") as f:
        blob = f.read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob


def test_header_is_plain_c():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "#include \"prifit_hip.h\"\nint main(void){return 0;}\n"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-x", "c", "-",
                        "-o", "/dev/null"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


def test_compat_aliases_resolve_to_backend(hiplib):
    """the reference's module paths (train_partseg_shapenet.py:219 importlib.import_module(args.model)) -> this backend"""
    import importlib
    import sys
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split(".")[0] in ("models", "src", "convex_loss")}
    try:
        from prifit_amd import compat
        names = compat.install()
        M = importlib.import_module("models.pointnet2_part_seg_msg")
        assert M.get_model.__module__.startswith("prifit_amd.") and hasattr(M, "get_loss")
        pu = importlib.import_module("models.pointnet_util")
        for n in ("square_distance", "index_points", "farthest_point_sample", "query_ball_point", "sample_and_group",
                  "sample_and_group_all", "PointNetSetAbstraction", "PointNetSetAbstractionMsg",
                  "PointNetFeaturePropagation"):
            assert hasattr(pu, n), n
        cl = importlib.import_module("convex_loss")
        for n in ("convex_loss", "entropy", "compute_sdf_ellipsoid", "compute_sdf_ellipsoids", "compute_sdf_ellipsoids_batch",
                  "compute_sdf_cuboid", "compute_sdf_cuboids", "compute_sdf_cuboid_batch", "compute_intersection_loss_volume_3",
                  "prune_points"):                                   # convex_loss.py:27,209,313-343,374-413,444-502
            assert callable(getattr(cl, n)), n
        eu = importlib.import_module("src.ellipsoid_utils")
        for n in ("guard_mean_shift", "clustering", "sample_from_pred_params", "sample_from_pred_params_cuboid",
                  "compute_approximate_ellipsoid_area"):             # src/ellipsoid_utils.py:9-214
            assert callable(getattr(eu, n)), n
        ef = importlib.import_module("src.ellipsoid_fitting")
        for n in ("weighted_ellipsoid_fitting", "weighted_ellipsoids_fitting", "weighted_ellipsoid_fitting_batch",
                  "principal_axis_ellipsoid"):                       # src/ellipsoid_fitting.py:19-141
            assert callable(getattr(ef, n)), n
        se = importlib.import_module("src.sample_ellipsoid").SampleEllipsoid
        assert callable(se.sample) and callable(se.sample_cuboid)    # src/sample_ellipsoid.py:17-96
        assert hasattr(importlib.import_module("src.mean_shift"), "MeanShift")
        assert len(names) >= 8
        net = M.get_model(50)
        keys = set(net.state_dict().keys())
        for k in ("sa1.conv_blocks.0.0.weight", "sa1.bn_blocks.2.2.running_var", "sa3.mlp_convs.2.bias",
                  "fp1.mlp_convs.0.weight", "conv1.weight", "bn1.running_mean", "conv2.bias", "extra_conv_emb.weight"):
            assert k in keys, k
        assert sum(p.numel() for p in net.parameters()) == 1757470  # model size
    finally:
        for k in list(sys.modules):
            if k.split(".")[0] in ("models", "src", "convex_loss"):
                del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
