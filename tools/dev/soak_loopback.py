#!/usr/bin/env python3
"""Soak of the counter-ordered pair / triple schedule with REAL peers: several ranks on one GPU, each a process of its own, exchanging through
libtm_hip's transport (csrc/tm_rccl.cpp) over the test-only loopback librccl (tests/loopback_rccl) -- random strips, owners per rank, reversed
interfaces, sweep counts; every run compared bit for bit with a single handle.   usage: soak_loopback.py [seed = 5] [cases = 16]"""
import json, os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LB = os.path.join(ROOT, "tests", "loopback_rccl", "libtm_loopback_rccl.so")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bad = 0
for case in range(cases):
    world = int(rng.integers(2, 5))
    bpr = int(rng.integers(1, 3)) if world <= 3 else 1
    topology = "junction" if (world == 4 and rng.random() < 0.3) else ("strip_rev" if rng.random() < 0.5 else "strip")
    big = rng.random() < 0.35
    ni, nj = (int(rng.integers(200, 420)), int(rng.integers(500, 1100))) if big else (int(rng.integers(16, 120)), int(rng.integers(16, 600)))
    if topology == "junction":
        ni, nj = int(rng.integers(20, 200)), int(rng.integers(20, 200))
    its = int(rng.integers(3, 40))
    triples = bool(rng.integers(0, 2))
    env = dict(os.environ, TM_RCCL_LIB=LB, HSA_ENABLE_IPC_MODE_LEGACY="0", TM_WORKER_BLOCKS_PER_RANK=str(bpr), TM_LOOPBACK_WAIT_S="60")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TM_PAIR_SYNC", "TM_TRIPLES_MIN_NODES"):
        env.pop(k, None)
    env["TM_TRIPLES_MIN_NODES"] = "1" if triples else "-1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.json")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(ROOT, "tests", "loopback_rccl", "worker.py"), "relax", topology, str(ni), str(nj), str(its), out], capture_output=True, text=True, env=env, cwd=ROOT)
        ok = r.returncode == 0 and os.path.exists(out) and json.load(open(out)).get("bit_identical_to_single_handle") is True
    bad += not ok
    print(f"case {case}: {world} ranks x {bpr} block(s), {topology} {ni} x {nj}, {its} sweeps, {'triples' if triples else 'pairs'}: {'ok' if ok else 'MISMATCH / FAILED ' + r.stderr[-300:]}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
