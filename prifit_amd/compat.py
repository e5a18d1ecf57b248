"""Drop-in aliasing: make the reference's module paths resolve to the MI355X backend.

    import prifit_amd.compat; prifit_amd.compat.install()
    MODEL = importlib.import_module("models.pointnet2_part_seg_msg")   # train_partseg_shapenet.py:219

After `install()` the names the reference's trainer / loss import -- `models.pointnet_util`,
`models.pointnet2_part_seg_msg`, `models.pretrain_pointnet2_part_seg_msg`, `convex_loss`,
`src.mean_shift`, `src.ellipsoid_fitting`, `src.ellipsoid_utils`, `src.fitting_utils`, `src.sample_ellipsoid`, `src.utils`,
`src.guard`, `src.VisUtils`, `data_utils.ShapeNetDataLoader`, `provider`, `testing` -- are this package's modules, and every
name `train_partseg_shapenet.py:5-28`, `testing.py:1-30` and `fitting.py:1-18` import from them resolves with the
reference's signature (tests/test_compat_imports.py holds the list).  What stays the reference's own: `args_parser` (its CLI)
and its third-party imports (tensorboard_logger, ipdb, tqdm, open3d, trimesh).
Nothing is imported from the reference tree."""
import importlib
import sys
import types

_ALIASES = {
    "models.pointnet_util": "prifit_amd.models.pointnet_util",
    "models.pointnet_utils": "prifit_amd.models.pointnet_util",
    "models.pointnet2_part_seg_msg": "prifit_amd.models.pointnet2_part_seg_msg",
    "models.pretrain_pointnet2_part_seg_msg": "prifit_amd.models.pretrain_pointnet2_part_seg_msg",
    "models.pointnet2_part_seg_ssg": "prifit_amd.models.pointnet2_part_seg_ssg",
    "models.pointnet2_cls_msg": "prifit_amd.models.pointnet2_cls_msg",
    "models.pointnet2_cls_ssg": "prifit_amd.models.pointnet2_cls_ssg",
    "models.pointnet2_sem_seg": "prifit_amd.models.pointnet2_sem_seg",
    "convex_loss": "prifit_amd.convex_loss",
    "testing": "prifit_amd.testing",
    "provider": "prifit_amd.provider",
    "data_utils.ShapeNetDataLoader": "prifit_amd.data",
    "src.mean_shift": "prifit_amd.src.mean_shift",
    "src.ellipsoid_fitting": "prifit_amd.src.ellipsoid_fitting",
    "src.ellipsoid_utils": "prifit_amd.src.ellipsoid_utils",
    "src.fitting_utils": "prifit_amd.src.fitting_utils",
    "src.sample_ellipsoid": "prifit_amd.src.sample_ellipsoid",
    "src.utils": "prifit_amd.src.utils",
    "src.VisUtils": "prifit_amd.src.utils",          # fitting.py:2 imports visualize_point_cloud from it
    "src.guard": "prifit_amd.src.guard",
    "src.dgcnn": "prifit_amd.src.dgcnn",
}


def install():
    for pkg in ("models", "src", "data_utils"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []  # namespace-like package
            sys.modules[pkg] = m
    for alias, target in _ALIASES.items():
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, mod)
    return sorted(_ALIASES)
