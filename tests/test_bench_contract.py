"""bench.py's one-line JSON contract and __graft_entry__.smoke(), on the GPU box (short runs)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-rows", "20000", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_bench_json_contract(mode):
    d = _run_bench(*(("--eager",) if mode == "eager" else ()))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "projections/s" and d["value"] > 1e7                      # BASELINE floor: >= 10 M proj/s
    assert abs(d["value"] - 1_000_000 * 30 / (d["ms_per_step"] * 30 * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or 0.9 * 72e6 < r["traffic"] < 1.5 * 72e6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "projections/s" and "sample" in c
    assert abs(d["mean_angle_error_delta_vs_ref_deg"]) < 1e-4                      # the parity half of the metric


def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
