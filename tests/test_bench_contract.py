"""bench.py and __graft_entry__ keep the driver's contract: one JSON line with the agreed fields (GPU), and the
argument parser / constants are sane (CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_constants_match_baseline():
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert (bench.ROWS, bench.COLS, bench.GROUP) == (65536, 1920, 128)
    assert "65536" in base["metric"] and "1920" in base["metric"] and "128" in base["metric"]
    assert bench.BYTES_PER_ELEM == 4 and bench.HBM_PEAK_GBS == 8000.0


def test_graft_entry_has_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
    assert "--offload-arch=gfx950" in g.HIP_FLAGS


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in res, key
    assert res["steps"] == 5 and res["warmup"] == 2 and res["n_gpus"] == 1 and res["vs_baseline"] is None
    assert res["unit"] == "Gelem/s" and res["dtype"] == "f16" and res["scaling"] == "weak" and "workload" in res["config"]
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.3 < rf["frac"] < 1.0, rf            # a 5-step run on a cold box is slower than the steady state, never absurd
    assert res["value"] > 400
