"""Test-only script (run in a fresh interpreter by tests/test_compat_dropin_cpu.py, build container only): the
REFERENCE's model files imported unmodified on top of ``graspbalance_amd.compat.install()``.

  level "L1": only the extension names are aliased (pointnet2._ext, pointnet2_batch_cuda, KNN._C): the reference's own
      pointnet2_utils.py / pointnet2_modules.py / ModifiedNetTools / knn_modules.py import and bind to this repo's
      extension shims; without a GPU every call must end in the reference's own error ("CPU not supported").
  level "L2": the python API names are aliased as well (pointnet2_utils, pointnet2_modules, pytorch_utils, group, ...):
      the reference's TrainModel/backbone.py and TrainModel/modules.py then build THEIR model classes out of THIS
      repo's layers; with the extension hooks bound to the CPU oracle their outputs must reproduce the fixtures made
      by the all-reference stack (tests/golden/g16_backbone.npz, g17_heads.npz).
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GB_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _third_party_stand_ins():
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    if "easydict" not in sys.modules:
        ed = types.ModuleType("easydict")

        class EasyDict(dict):
            __getattr__ = dict.__getitem__
            __setattr__ = dict.__setitem__
        ed.EasyDict = EasyDict
        sys.modules["easydict"] = ed


def level1():
    from graspbalance_amd import compat
    done = compat.install(names=("pointnet2", "pointnet2._ext", "pointnet2_batch_cuda", "KNN", "KNN._C"), override=True)
    assert set(done) == {"pointnet2", "pointnet2._ext", "pointnet2_batch_cuda", "KNN", "KNN._C"}, done
    _third_party_stand_ins()
    for sub in ("KNN", "ModifiedNetTools", "PointNet", "TrainModel", ""):
        sys.path.insert(0, os.path.join(REF, sub))
    import pointnet2_utils as ref_pu          # the reference's file: `import pointnet2._ext as _ext` (:20-28)
    import pointnet2_modules as ref_pm        # noqa: F401
    import group as ref_group                 # `import pointnet2_batch_cuda as pointnet2_cuda` (group.py:13)
    import knn_modules as ref_knn             # `from KNN import _C` (knn_modules.py:6)
    import backbone as ref_backbone
    import modules as ref_modules             # noqa: F401
    import graspbalance_amd.pointnet2._ext as own_ext
    import graspbalance_amd.pointnet2_batch_cuda as own_pb
    assert ref_pu.__file__.startswith(REF) and ref_group.__file__.startswith(REF) and ref_knn.__file__.startswith(REF)
    assert ref_pu._ext is own_ext and ref_group.pointnet2_cuda is own_pb
    net = ref_backbone.Pointnet2Backbone()
    assert sum(p.numel() for p in net.parameters()) == 641856
    for call in (lambda: net(torch.rand(1, 4096, 3)),
                 lambda: ref_pu.ball_query(0.1, 8, torch.rand(1, 64, 3), torch.rand(1, 8, 3)),
                 lambda: ref_knn.myknn(torch.rand(1, 3, 16), torch.rand(1, 3, 4))):
        try:
            call()
        except RuntimeError as e:
            assert "CPU not supported" in str(e) or "must be a CUDA tensor" in str(e), e
        else:
            raise AssertionError("a CPU call through the drop-in extension must raise like the reference's")
    print("L1 ok")


def level2():
    from graspbalance_amd import compat
    names = ("pointnet2", "pointnet2._ext", "pointnet2_batch_cuda", "pointnet2_utils", "pointnet2_modules",
             "pytorch_utils", "knn_modules", "KNN", "KNN._C", "group", "subsample", "upsampling", "conv", "norm",
             "activation", "loss_utils")
    compat.install(names=names, override=True)
    _third_party_stand_ins()
    from tests import cpu_backend
    cpu_backend.install()  # no GPU here: the extension hooks of THIS repo's python layers -> CPU oracle
    sys.path.insert(0, os.path.join(REF, "TrainModel"))
    import backbone as ref_backbone
    import modules as ref_modules
    import graspbalance_amd.pointnet2_modules as own_pm
    assert ref_backbone.__file__.startswith(REF) and ref_modules.__file__.startswith(REF)
    assert ref_backbone.PointnetSAModuleVotes is own_pm.PointnetSAModuleVotes
    from tests.golden import make_golden_r2 as mk
    from tests.seeded import assert_errors, fill_by_key
    from tests import test_reference_fixtures_cpu as cases
    net = fill_by_key(ref_backbone.Pointnet2Backbone(), seed=16)
    errs = cases.run_backbone_case(net, mk.g16_cloud(), np.load(os.path.join(ROOT, "tests", "golden", "g16_backbone.npz")))
    assert_errors(errs, {"grad/": 2e-5}, 2e-6)
    import graspbalance_amd.modules as own_modules
    saved = {k: getattr(own_modules, k) for k in ("GraspableDetection", "GraspWidthGrouping",
                                                  "GraspPoseParametersDetection", "ToleranceNet")}
    try:  # run_heads_case builds graspbalance_amd.modules.*: hand it the REFERENCE's head classes instead
        for k in saved:
            setattr(own_modules, k, getattr(ref_modules, k))
        errs = cases.run_heads_case("cpu", np.load(os.path.join(ROOT, "tests", "golden", "g17_heads.npz")))
    finally:
        for k, v in saved.items():
            setattr(own_modules, k, v)
    assert_errors(errs, {}, 2e-6)
    print("L2 ok")


if __name__ == "__main__":
    {"L1": level1, "L2": level2}[sys.argv[1]]()
