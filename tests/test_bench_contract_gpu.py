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
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra-configs"])
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
    # round 5: honest side figures - launch-by-launch execution as shipped (not the instrumented leg), the step on
    # changing data, the memory the capturable step costs, and where `traffic` comes from
    assert d["ms_per_step_eager"] > 0 and d["ms_per_step_instrumented_leg"] > 0 and d["ms_per_step_changing_data"] > 0
    assert 1.0 < d["max_memory_allocated_gb"] < 288.0
    src = r["traffic_source"]
    assert src["file"].startswith("profiles/") and (r["traffic"] is None) == (not src["valid"])
    # round 6: a kernel that issues six bf16 MFMAs per fp32 product is priced against the roof that
    # binds it - the bf16 matrix cores on EXECUTED flop or HBM on its bytes - not against the fp32 MFMA peak
    split = [d[k] for k in ("roofline", "roofline_gemm2", "roofline_gemm3", "roofline_gemm4", "roofline_gemm5")
             if d.get(k) and "fp32_equivalent_frac" in d[k]]
    assert len(split) == 2, [x["kernel"][:24] for x in split]     # gemm_rs_kernel and wgrad_direct_kernel
    for x in split:
        assert abs(x["frac"] - max(x["mfma_executed_frac"], x["hbm_frac"])) < 1e-3 and 0 < x["frac"] < 1
        assert abs(x["frac"] - x["achieved"] / x["peak"]) < 2e-3 and x["unit"] in ("GB/s", "TFLOP/s")
        assert abs(x["mfma_executed_frac"] - 6 * x["tflops_fp32_equivalent"] / 2500.0) < 1e-3
        assert x["bound"] == ("hbm" if x["hbm_frac"] >= x["mfma_executed_frac"] else "mfma")


def test_bench_line_carries_the_other_two_configurations():
    """the default command's line also reports BASELINE configs[2] (eval forward + decode) and
    configs[4] (B = 8 x 50 000, bf16 contractions) - short legs in child processes after the timed region."""
    import os
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GB_FORCE_DIST")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    inf, st = d["configs"]["infer"], d["configs"]["stress"]
    assert "error" not in inf and "error" not in st, (inf, st)
    assert inf["ms_per_step"] > 0 and "configs[2]" in inf["config"]["workload"] and inf["roofline"]["bound"] == "hbm"
    assert st["ms_per_step"] > 0 and "configs[4]" in st["config"]["workload"] and st["dtype"] == "bf16"
    assert abs(st["value"] - 8 / (st["ms_per_step"] * 1e-3)) / st["value"] < 1e-3
