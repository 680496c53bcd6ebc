"""bench.py and __graft_entry__.smoke() as the driver runs them: one JSON line with the contract's keys."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def run_bench(*args):
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), *args], capture_output=True, text=True, timeout=600,
                         cwd=str(REPO))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must be one JSON line, got {len(lines)}"
    return json.loads(lines[0])


def test_default_bench_line():
    d = run_bench("--steps", "30", "--warmup", "5")
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], kind), (key, d[key])
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this workload
    assert "scaling" in d and d["scaling"] is None        # one GPU: neither weak nor strong
    assert (d["n_gpus"], d["steps"], d["warmup"], d["dtype"], d["higher_is_better"]) == (1, 30, 5, "f16", True)
    assert "workload" in d["config"] and "nips" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 8000.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "GFLOP/s" and c["sample"]
    assert d["parity_mismatches_vs_cpu"] == 0
    # value = 2 * nnz * K / time
    assert abs(d["value"] - 2 * 746316 * 128 / (d["ms_per_step"] * 1e6)) / d["value"] < 1e-3


def test_real_matrix_workload_carries_the_published_number():
    d = run_bench("--workload", "wathen100_k128", "--steps", "20", "--warmup", "3", "--no-cpu-baseline")
    assert d["published_reference"]["gflops"] == 2281.83
    assert abs(d["vs_baseline"] - d["value"] / 2281.83) < 1e-2
    assert "SuiteSparse" in d["data"]


def test_smoke_entry():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True,
                         timeout=600, cwd=str(REPO))
    assert out.returncode == 0, out.stderr[-2000:]
