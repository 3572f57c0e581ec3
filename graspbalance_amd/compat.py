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


def install(names=None, override=False):
    """Register the aliases (all, or the given subset).  Existing entries are kept unless ``override``."""
    done = []
    for alias, target in _ALIASES.items():
        if names is not None and alias not in names:
            continue
        if alias in sys.modules and not override:
            continue
        sys.modules[alias] = importlib.import_module(target)
        done.append(alias)
    return done


class _KNNPackage:
    """``from KNN import _C`` (KNN/knn_modules.py:6): ``_C.knn(ref, query, idx)``."""

    class _C:
        @staticmethod
        def knn(ref, query, idx):
            from . import knn_modules
            return knn_modules.knn(ref, query, idx)
