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

    b = bench.cpu_baseline(129, budget_s=0.3, with_t106=False)
    assert BASELINE <= set(b) and b["kind"] == "port" and b["cores"] == 1 and b["value"] > 0
    # the reference's path stage by stage (fill, BiCGStab-diagonal, GMRES+ILU0, residual + copy-back), sizes stated
    st = b["stages"]
    assert {"tfi", "init", "fill", "bicgstab_diag", "gmres30_ilu0", "residual_copyback"} <= set(st)
    assert st["bicgstab_diag"]["iterations"] == 3 and st["gmres30_ilu0"]["iterations"] == 4
    assert all(st[k]["seconds"] > 0 for k in ("fill", "bicgstab_diag", "gmres30_ilu0", "residual_copyback"))
    m = b["mirror_sweep"]
    assert m["cores"] == 1 and m["value"] > 0 and m["all_threads"]["cores"] >= 1 and m["all_threads"]["value"] > 0


def test_t106_json_as_written_cpu_leg():
    sys.path.insert(0, ROOT)
    import bench

    t = bench.t106_cpu()   # BASELINE configs[0]: 10 iterations, GMRES + ILU0, white
    assert t["outer_iterations"] == 10 and t["nodes"] == 25118 and t["cpu_seconds"] > 0


def test_self_launch_command_line(monkeypatch):
    # `python bench.py --gpus N` without a launcher starts N ranks as a CHILD process and relays rank 0's JSON line
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    class R:
        returncode = 0
        stdout = b'noise\n{"metric": "x", "n_gpus": 4}\n'

    def fake_run(cmd, stdout=None, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


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
    assert 0 < j["roofline"]["frac"] <= 1.0 and j["roofline"]["launches_timed"] >= 1   # a bandwidth fraction, never above the peak


@pytest.mark.gpu
@pytest.mark.parametrize("pair_sync", ["counters", "events", "triples"])
def test_bench_gpus4_config4_rehearsal_on_one_gpu_is_bit_identical_to_one_handle(pair_sync):
    # `python bench.py --gpus 4 --config 4` launches its own ranks; rehearsed on ONE GPU with the gloo transport (halo rows staged
    # through the host; RCCL refuses two ranks on one device): strong scaling of the 8-block strip, 2 blocks per rank, and after
    # 4+21 sweeps every rank's blocks equal the single-handle run of the whole strip bit for bit.  4 ranks + this process <= 6.
    # Both forms of the cross-queue ordering inside a sweep pair: one-wave signal / wait kernels on device-memory counters (the
    # default for one multi-rank handle per process) and hipEvent record / wait (TM_PAIR_SYNC=events).
    env = dict(os.environ, TM_BENCH_SAME_DEVICE="1", TM_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("TM_PAIR_SYNC", None)
    if pair_sync == "events":
        env["TM_PAIR_SYNC"] = "events"
    env["TM_TRIPLES_MIN_NODES"] = "-1"   # the PAIR schedule (since round 4 the default is triples at every size)
    if pair_sync == "triples":   # sweep triples across ranks (depth-3 halo, one exchange per triple): the threshold lowered to these 96^2 blocks
        env["TM_TRIPLES_MIN_NODES"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "4", "--size", "96", "--steps", "21", "--warmup", "4",
                        "--verify", "--transport", "torch", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and j["config"]["nodes_total"] == 8 * 96 * 96 and j["config"]["nodes_per_gpu"] == 2 * 96 * 96
    assert j["config"]["verified_against_single_handle"] is True
    assert 0 < j["roofline"]["frac"] <= 1.0
