"""libtm_hip's own transport (csrc/tm_rccl.cpp: grouped ncclRecv / ncclSend per neighbouring rank, ncclAllReduce) with N > 1 ranks.

Real RCCL refuses two ranks on one device and this pool hands out one GPU, so the ranks share it and tm_rccl.cpp dlopens a TEST-ONLY
stand-in for librccl (tests/loopback_rccl: mailboxes in hipIpc device memory, every copy and wait enqueued on the stream ncclSend /
ncclRecv name) by path -- the same `librccl_path` argument a site uses for its own RCCL build.  Everything above that dlopen is the
product: the peer tables of tm_rccl_hooks, the exchange issued on the chain's stream, the counter-ordered pair / triple schedule.
What is asserted: after W + K sweeps every rank's blocks equal a single-handle run of the whole mesh BIT FOR BIT."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LB_DIR = os.path.join(ROOT, "tests", "loopback_rccl")
LB = os.path.join(LB_DIR, "libtm_loopback_rccl.so")
NEEDED = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclAllReduce", "ncclGetErrorString"]


def _build():
    src = os.path.join(LB_DIR, "loopback_rccl.hip")
    if os.path.exists(LB) and os.path.getmtime(LB) >= os.path.getmtime(src):
        return True
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return os.path.exists(LB)
    return subprocess.run(["make", "-C", LB_DIR, "-s"], capture_output=True).returncode == 0 and os.path.exists(LB)


def test_loopback_library_exports_what_tm_rccl_resolves():
    # CPU: the stand-in builds for gfx950 and offers every symbol csrc/tm_rccl.cpp looks up with dlsym -- the list is read from that file
    assert _build(), "tests/loopback_rccl does not build"
    src = open(os.path.join(ROOT, "turbomesh_amd", "csrc", "tm_rccl.cpp")).read()
    import re

    wanted = sorted(set(re.findall(r'sym\("(nccl\w+)"\)', src)))
    assert wanted == sorted(NEEDED)
    out = subprocess.run(["nm", "-D", "--defined-only", LB], capture_output=True, text=True).stdout
    have = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert set(wanted) <= have


def test_the_product_never_names_the_loopback_library():
    for base, _, files in os.walk(os.path.join(ROOT, "turbomesh_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(base, f), errors="replace").read().lower()
                assert "loopback" not in text, f"{f} mentions the test-only transport"
    assert "loopback" not in open(os.path.join(ROOT, "include", "tm_hip.h")).read().lower()


def _env(**extra):
    env = dict(os.environ, TM_BENCH_SAME_DEVICE="1", TM_BENCH_BACKEND="gloo", TM_RCCL_LIB=LB, HSA_ENABLE_IPC_MODE_LEGACY="0", TM_LOOPBACK_WAIT_S="60")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TM_PAIR_SYNC", "TM_TRIPLES_MIN_NODES"):
        env.pop(k, None)
    env.update(extra)
    return env


def _bench(args, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--verify", "--transport", "rccl", "--no-cpu-baseline", "--settle-ms", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert "RCCL p2p issued by libtm_hip" in j["config"]["workload"] and "libtm_loopback_rccl.so" in j["config"]["workload"], j["config"]["workload"]
    return j


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["pairs", "triples", "events"])
def test_bench_gpus4_config4_through_tm_rccl_is_bit_identical_to_one_handle(schedule):
    # BASELINE configs[3] in small: the 8-block strip over 4 ranks (2 blocks per rank), exchanges issued by tm_rccl.cpp on the chain's stream.
    # pairs: depth-2 halo, one exchange per sweep pair, counters; triples: depth-3 halo, one exchange per triple, announce-and-wait kernels;
    # events: the same with hipEvent ordering.  4 ranks + this process <= 6 processes on the card.
    assert _build()
    extra = {"events": {"TM_PAIR_SYNC": "events", "TM_TRIPLES_MIN_NODES": "1"}, "triples": {"TM_TRIPLES_MIN_NODES": "1"}, "pairs": {"TM_TRIPLES_MIN_NODES": "-1"}}[schedule]
    j = _bench(["--gpus", "4", "--config", "4", "--size", "96", "--steps", "21", "--warmup", "4"], _env(**extra))
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and j["config"]["nodes_per_gpu"] == 2 * 96 * 96
    assert j["config"]["verified_against_single_handle"] is True
    assert j["config"]["pair_sync"] == ("events" if schedule == "events" else "counters")


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,size", [(2, 160), (3, 128)])
def test_bench_weak_scaling_line_through_tm_rccl_triples(gpus, size):
    # the default bench line at N > 1 (a strip of N blocks, one per rank; rank 1 of 3 has neighbours on both sides), triples threshold lowered
    assert _build()
    j = _bench(["--gpus", str(gpus), "--size", str(size), "--steps", "18", "--warmup", "3"], _env(TM_TRIPLES_MIN_NODES="1"))
    assert j["n_gpus"] == gpus and j["scaling"] == "weak" and j["config"]["verified_against_single_handle"] is True


def _worker(world, args, env, tmp_path):
    out = tmp_path / "result.json"
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(LB_DIR, "worker.py")] + [str(a) for a in args] + [str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:] + r.stderr[-4000:])
    return json.load(open(out))


@pytest.mark.gpu
@pytest.mark.parametrize("topology,world,ni,nj,its", [("strip_rev", 3, 48, 72, 13), ("junction", 4, 40, 44, 11)])
def test_relax_sweeps_over_reversed_interfaces_and_the_junction_mesh(topology, world, ni, nj, its, tmp_path):
    # reversed ranges (ascending on one side, descending on the other) and the 2 x 2 junction mesh (every rank has two neighbours and the
    # centre node's row reads three remote blocks): pairs first, then triples with the threshold lowered
    assert _build()
    for extra in ({"TM_TRIPLES_MIN_NODES": "-1"}, {"TM_TRIPLES_MIN_NODES": "1"}):
        res = _worker(world, ["relax", topology, ni, nj, its], _env(**extra), tmp_path)
        assert res["bit_identical_to_single_handle"] is True, res


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["krylov", "gmres", "mg"])
def test_krylov_path_over_tm_rccl_allreduce(mode, tmp_path):
    # Picard + BiCGStab (and GMRES(30): an all-reduce per Gram-Schmidt inner product) on two ranks: halo exchange per operator application +
    # ncclAllReduce of the reduction scalars, both issued by tm_rccl.cpp.  Two summands commute, so the run is bit-identical to the
    # torch.distributed hooks; a single handle sums in another order: <= 1e-10 rms.  "mg": the multigrid-preconditioned solve, whose perimeter step
    # exchanges the corrections inside every preconditioner application (Smoother::precondition).
    assert _build()
    res = _worker(2, [mode, "strip", 40, 56, 2], _env(), tmp_path)
    assert res["bit_identical_to_torch_hooks"] is True, res
    assert res["rms_vs_single_handle"] <= 1e-10, res
    if mode == "krylov":   # blocks that count as large: the transport's tables are built for the handle's options (tm_rccl_hooks_for: depth 2, like the torch hooks' plan)
        res = _worker(2, [mode, "strip", 40, 56, 2], _env(TM_TRIPLES_MIN_NODES="1"), tmp_path)
        assert res["bit_identical_to_torch_hooks"] is True and res["rms_vs_single_handle"] <= 1e-10, res
