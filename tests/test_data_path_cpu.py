"""CPU: the numpy restatement of the host data path (oracle/data_path.py) against a fixture produced by the
reference's own data_utils.py (tests/golden/make_golden_r2.py g18)."""
import hashlib

import numpy as np
import torch

from oracle import data_path
from tests.seeded import check_summary


def test_oracle_data_path_matches_reference(golden):
    g = golden.load("g18_data_path")
    for seed in (0, 1):
        depth, seg, cam, trans = data_path.synthetic_frame(seed)
        out = data_path.frame_to_cloud(depth, seg, *cam, trans=trans, outlier=0.02)
        check_summary(g, "f%d/cloud" % seed, torch.from_numpy(out["cloud"]), 0.0)
        check_summary(g, "f%d/cloud_masked" % seed, torch.from_numpy(out["cloud_masked"]), 0.0)
        assert int(out["mask"].sum()) == int(g["f%d_mask_count" % seed])
        assert hashlib.sha256(np.packbits(out["mask"]).tobytes()).digest() == g["f%d_mask_sha256" % seed].tobytes()
        assert hashlib.sha256(np.packbits(out["workspace_mask"]).tobytes()).digest() == g["f%d_workspace_mask_sha256" % seed].tobytes()
        assert int(out["seg_masked"].astype(np.int64).sum()) == int(g["f%d_seg_masked_sum" % seed])
