"""CPU, build container only (the reference tree does not travel): the drop-in boundary of INTEGRATION.md exercised
with the reference's own files - see tests/dropin_check.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GB_REFERENCE", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent (GPU box)")


def test_install_registers_every_alias():
    """compat.install() in a fresh interpreter: every alias resolves, incl. ``from KNN import _C``."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from graspbalance_amd import compat\n"
            "done = compat.install()\n"
            "assert set(done) >= set(compat._ALIASES) | {'KNN', 'KNN._C'}, done\n"
            "import pointnet2._ext as e, pointnet2_batch_cuda as pb, pointnet2_utils, pointnet2_modules, pytorch_utils\n"
            "from KNN import _C\n"
            "import KNN._C as c2\n"
            "assert _C is c2 and callable(_C.knn)\n"
            "from knn_modules import myknn\n"
            "assert all(hasattr(e, n) for n in ('furthest_point_sampling', 'ball_query', 'cylinder_query', 'group_points',"
            " 'three_nn', 'three_interpolate', 'gather_points', 'gather_points_grad', 'group_points_grad', 'three_interpolate_grad'))\n"
            "assert all(hasattr(pb, n) for n in ('ball_query_wrapper', 'group_points_wrapper', 'furthest_point_sampling_wrapper'))\n"
            "assert compat.install() == []\n" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])


@needs_reference
@pytest.mark.parametrize("level", ["L1", "L2"])
def test_reference_model_files_run_on_the_drop_in(level):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_check.py"), level],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert level + " ok" in out.stdout
