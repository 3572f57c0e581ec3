"""GPU: bench.py prints ONE JSON line with the fields the driver's contract names (1 timed step, run in-process)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GB_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3  # clouds per second of the whole job
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    for k in ("roofline_fps", "roofline_ball", "roofline_fps_ball", "roofline_cyl"):
        assert d[k] is not None and 0.0 < d[k]["frac"] < 1.0, (k, d[k])
    assert d["roofline_ball"]["bound"] == "valu" and 0 < d["roofline_ball"]["scanned_frac_of_full"] <= 1
    assert d["ranks_seen"] == 1
