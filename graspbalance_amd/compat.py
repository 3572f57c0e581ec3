"""Make the reference's import names resolve to this implementation.

The reference puts its source folders on ``sys.path`` and imports top-level modules
(``import pointnet2._ext as _ext`` pointnet2_utils.py:20, ``import pointnet2_batch_cuda``
group.py:13, ``import pytorch_utils as pt_utils``, ``from conv import create_convblock1d`` drp.py:14,
``from knn_modules import myknn`` label_generation.py:10 ...).  ``install()`` registers the same names
in ``sys.modules`` so those import statements — in the reference's graspbalance.py / train.py or in
user code — bind to the MI355X implementation without edits.
"""
import importlib
import sys

_ALIASES = {
    "pointnet2": "graspbalance_amd.pointnet2",
    "pointnet2._ext": "graspbalance_amd.pointnet2._ext",
    "pointnet2_batch_cuda": "graspbalance_amd.pointnet2_batch_cuda",
    "pointnet2_utils": "graspbalance_amd.pointnet2_utils",
    "pointnet2_modules": "graspbalance_amd.pointnet2_modules",
    "pytorch_utils": "graspbalance_amd.pytorch_utils",
    "knn_modules": "graspbalance_amd.knn_modules",
    "group": "graspbalance_amd.modified_net_tools.group",
    "subsample": "graspbalance_amd.modified_net_tools.subsample",
    "upsampling": "graspbalance_amd.modified_net_tools.upsampling",
    "conv": "graspbalance_amd.modified_net_tools.conv",
    "norm": "graspbalance_amd.modified_net_tools.norm",
    "activation": "graspbalance_amd.modified_net_tools.activation",
    "backbone": "graspbalance_amd.backbone",
    "drp": "graspbalance_amd.drp",
    "loss_utils": "graspbalance_amd.loss_utils",
    "label_generation": "graspbalance_amd.label_generation",
}


def _knn_package():
    """``from KNN import _C`` (KNN/knn_modules.py:6) and ``import KNN._C``: ``_C.knn(ref, query, idx)``."""
    import types
    from . import knn_modules
    pkg = types.ModuleType("KNN")
    pkg.__path__ = []  # a package, so that ``import KNN._C`` resolves through sys.modules
    ext = types.ModuleType("KNN._C")
    ext.knn = lambda ref, query, idx: knn_modules.knn(ref, query, idx)  # late-bound: follows knn_modules.knn
    pkg._C = ext
    return {"KNN": pkg, "KNN._C": ext}


def install(names=None, override=False):
    """Register the aliases (all, or the given subset).  Existing entries are kept unless ``override``."""
    done = []
    targets = {alias: (lambda t=target: importlib.import_module(t)) for alias, target in _ALIASES.items()}
    knn = None
    for alias in ("KNN", "KNN._C"):
        targets[alias] = None
    for alias, load in targets.items():
        if names is not None and alias not in names:
            continue
        if alias in sys.modules and not override:
            continue
        if load is None:
            knn = knn or _knn_package()
            sys.modules[alias] = knn[alias]
        else:
            sys.modules[alias] = load()
        done.append(alias)
    return done
