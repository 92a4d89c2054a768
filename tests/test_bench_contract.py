"""The driver's contract with bench.py: ONE JSON line on stdout with the agreed keys (task description, "Measurement")."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
BASELINE = {"value", "unit", "cores", "kind", "sample"}


def test_cpu_baseline_leg_runs_on_the_host_alone():
    # the only place outside tests/ and smoke() that may use the oracle; bounded sample, no GPU involved
    sys.path.insert(0, ROOT)
    import bench

    b = bench.cpu_baseline(129, budget_s=0.3)
    assert BASELINE <= set(b) and b["kind"] == "port" and b["cores"] == 1 and b["value"] > 0
    assert b["all_threads"]["cores"] >= 1 and b["all_threads"]["value"] > 0


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "512", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-solve"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert KEYS <= set(j) and ROOFLINE <= set(j["roofline"])
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["dtype"] == "f64" and j["data"] == "synthetic" and j["vs_baseline"] is None and "workload" in j["config"]
    assert j["value"] > 0 and j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0
