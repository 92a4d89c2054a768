#!/usr/bin/env python3
"""bench.py -- nodes smoothed/sec + achieved HBM GB/s of the elliptic sweep (BASELINE.json metric).

A step = ONE fused elliptic sweep of the workload (SURVEY.md 8d "unit of work"): the
frozen-coefficient 9-point Winslow operator applied to both coordinate components of every
node, with Jacobi scaling, the relaxation update, the perimeter (constraint / interface)
rows and the residual-norm partial reduction -- i.e. one outer iteration of the `hip` solver
in TM_INNER_RELAX mode, run through the C-ABI handle with the coordinates resident in HBM.

  --config 2 (default)  N = 1: BASELINE configs[1] -- single synthetic 4096 x 4096 block, TFI
                        seeded on the GPU.  N > 1: weak scaling -- a strip of N such blocks stacked in i, one per GPU,
                        coupled by interface rows exchanged point-to-point (RCCL over xGMI), once per three sweeps.  (A lone
                        block with fixed walls needs no perimeter-row passes beside its interior pass: the N = 1 value is not quite
                        the per-GPU ceiling of the N > 1 runs -- 37 against 40 us per sweep -- see DESIGN.md section 6.)
  --config 4            BASELINE configs[3], STRONG scaling: 8 coupled blocks of 2048^2, 8/N blocks per GPU.
  --config 5            BASELINE configs[4]: independent 2048^2 slices, 8 per GPU ("replicas only": no communication).

`python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed.run as a child process)
when it is not already running under a launcher.  Prints ONE JSON line on rank 0."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)
BYTES_PER_NODE = 32.0         # compulsory traffic of one pass over a Laplace field: read one double2, write one double2
PROFILE_EVERY = 4             # a hipEvent pair brackets every 4th launch of the dominant kernel inside the timed region


def kernels_hash():
    """Identifies the kernel source a PMC measurement belongs to (profiles/traffic.json carries the hash it was taken with)."""
    with open(os.path.join(ROOT, "turbomesh_amd", "csrc", "tm_kernels.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------
# cpu_baseline: the oracle (a C++ restatement of the reference's Zig; the reference itself cannot be built here) timed
# on this box's host cores, bounded samples.  Only this function touches oracle/.
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(n, budget_s=15.0, with_t106=True):
    """value = the reference's own per-outer-iteration path (CSR fill + BiCGStab with the diagonal preconditioner + residual +
    copy-back, smooth.zig:104-154, BiCGStab.zig:279-370) in node-operator-applications per second, 1 thread, on a
    sub-block of the workload; the stages, GMRES(30)+ILU(0) (what examples/T106/T106.json selects), the T106 JSON as written
    and the matrix-free mirror sweep (the same unit of work as the GPU step) are reported as structured keys beside it."""
    import numpy as np

    from oracle import oracle
    from turbomesh_amd import configs

    def tfi_cpu(i_min, i_max, j_min, j_max):
        return configs.block_from_array(oracle.tfi_block(i_min.points, i_max.points, j_min.points, j_max.points, i_min.clustering,
                                                          i_max.clustering, j_min.clustering, j_max.clustering))

    t0 = time.perf_counter()
    mesh = configs.single_block(n, n, tfi=tfi_cpu)
    t_tfi = time.perf_counter() - t0          # includes building the four edges (numpy), dominated by the TFI loop
    xy = mesh.blocks[0].points.data

    # ---- the reference's path, stage by stage, on a sub-block (the CSR + GMRES basis of 4096^2 need ~8 GB and minutes)
    m = min(n, 2048 if budget_s >= 5 else 257)
    sub = np.ascontiguousarray(xy[:m, :m]).copy()
    bi, gi = (8, 10) if budget_s >= 5 else (3, 4)
    st = oracle.time_reference_path(sub, bi, gi)
    nodes = float(m * m)
    bic_apps = 2 * st["bicgstab_iterations"] + 1                 # 2 mat-vecs per iteration + the initial residual
    gm_apps = st["gmres_iterations"] + 1
    path_s = st["fill_s"] + st["bicgstab_diag_s"] + st["residual_copyback_s"]
    stages = {
        "sample_block": f"{m}x{m} sub-block of the {n}x{n} workload, x-system (BiCGStab) / y-system (GMRES), 1 thread",
        "tfi": {"seconds": t_tfi, "nodes_per_s": n * n / t_tfi, "what": f"tfi.zig:112-208 on the {n}x{n} block"},
        "init": {"seconds": st["init_s"], "what": "RowCompressedMatrixSystem2d.init (pattern, row kinds, static rows), once per smooth.mesh call"},
        "fill": {"seconds": st["fill_s"], "nodes_per_s": nodes / st["fill_s"], "what": "system.fill: 9 coefficients per row, smooth.zig:923-1113"},
        "bicgstab_diag": {"seconds": st["bicgstab_diag_s"], "iterations": st["bicgstab_iterations"],
                          "node_matvecs_per_s": nodes * bic_apps / st["bicgstab_diag_s"], "what": "BiCGStab.zig:279-370, one component"},
        "gmres30_ilu0": {"seconds": st["gmres30_ilu0_s"], "iterations": st["gmres_iterations"], "ilu0_factor_seconds": st["ilu0_factor_s"],
                         "node_matvecs_per_s": nodes * gm_apps / st["gmres30_ilu0_s"],
                         "what": "GMRES.zig:300-423 + ILU(0) GMRES.zig:199-298, one component (the solver of examples/T106/T106.json)"},
        "residual_copyback": {"seconds": st["residual_copyback_s"], "nodes_per_s": nodes / st["residual_copyback_s"], "what": "smooth.zig:112-153"},
    }

    # ---- the matrix-free mirror sweep: the GPU step's own unit of work on the CPU (NOT something the reference executes)
    t1 = oracle.time_relax_sweeps(xy, 1)
    sweeps = max(1, min(512, int(0.4 * budget_s / max(t1, 1e-3))))
    t = oracle.time_relax_sweeps(xy, sweeps)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads, 64))
    t1m = oracle.time_relax_sweeps_mt(xy, 2, threads)
    sweeps_m = max(2, min(2048, int(0.3 * budget_s / max(t1m / 2, 1e-4))))
    tm = oracle.time_relax_sweeps_mt(xy, sweeps_m, threads)
    mirror = {"value": n * n * sweeps / t, "unit": "nodes/s", "cores": 1,
              "sample": f"{sweeps} matrix-free Jacobi elliptic sweeps of the {n}x{n} block (oracle/orc_mirror.cpp, g++ -O2 -ffp-contract=off, {t:.1f} s)",
              "all_threads": {"value": n * n * sweeps_m / tm, "cores": threads, "sample": f"{sweeps_m} sweeps, rows split over {threads} threads ({tm:.1f} s)"}}

    out = {
        "value": nodes * bic_apps / path_s, "unit": "nodes/s", "cores": 1, "kind": "port",
        "sample": (f"one outer iteration of the reference path on a {m}x{m} sub-block: fill + {st['bicgstab_iterations']} BiCGStab(diagonal) "
                   f"iterations ({bic_apps} operator applications) + residual/copy-back, {path_s:.1f} s; C++ restatement (g++ -O2 "
                   "-ffp-contract=off), the Zig reference cannot be built in this image"),
        "stages": stages, "mirror_sweep": mirror, "host_cpus": os.cpu_count(),
    }
    if with_t106:
        out["t106_json_as_written"] = t106_cpu()
    return out


def _t106_load(tfi):
    from turbomesh_amd.input import Input

    gold = os.path.join(ROOT, "tests", "golden")
    inp = Input.parse(open(os.path.join(gold, "examples", "T106", "T106.json")).read())
    return inp, inp.template.run(inp.geometry(gold), tfi=tfi)


def t106_cpu():
    """BASELINE configs[0]: examples/T106/T106.json as written (8 blocks, 25 118 nodes, 10 iterations, GMRES+ILU0, white) on the oracle."""
    import numpy as np

    from oracle import oracle
    from turbomesh_amd import configs

    def tfi_cpu(i_min, i_max, j_min, j_max):
        return configs.block_from_array(oracle.tfi_block(i_min.points, i_max.points, j_min.points, j_max.points, i_min.clustering,
                                                          i_max.clustering, j_min.clustering, j_max.clustering))

    class OM:
        def __init__(self, mesh):
            self.blocks = [np.array(b.points.data, dtype=np.float64, order="C", copy=True) for b in mesh.blocks]
            self.connections = [((c.ranges[0].block, int(c.ranges[0].side), c.ranges[0].start, c.ranges[0].end),
                                 (c.ranges[1].block, int(c.ranges[1].side), c.ranges[1].start, c.ranges[1].end),
                                 None if c.periodicity is None else tuple(c.periodicity)) for c in mesh.connections]
            self.bcs = [((b.range.block, int(b.range.side), b.range.start, b.range.end), int(b.kind)) for b in mesh.boundary_conditions]

    inp, mesh = _t106_load(tfi_cpu)
    w = inp.wall_control_function.white
    om = OM(mesh)
    t0 = time.perf_counter()
    st = oracle.smooth_mesh(om, inp.iterations, solver=oracle.SOLVER_GMRES, preconditioner=oracle.PRECOND_ILU0, control=("white", w.ds_target, w.theta_target))
    dt = time.perf_counter() - t0
    return {"cpu_seconds": dt, "outer_iterations": int(st.outer_iterations), "inner_iterations": int(st.inner_iterations),
            "nodes": int(sum(b.shape[0] * b.shape[1] for b in om.blocks)), "solver": "gmres(30) + ilu0, white control function (the JSON as written)",
            "nodes_per_s": sum(b.shape[0] * b.shape[1] for b in om.blocks) * inp.iterations / dt}


def t106_gpu(smooth, solver, wcf):
    """The same job on the GPU: TFI of the 8 blocks + 10 Picard iterations with the white control function, hip solver."""
    t0 = time.perf_counter()
    inp, mesh = _t106_load(None)
    t_tfi = time.perf_counter() - t0
    w = inp.wall_control_function.white
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-8, max_inner=20000), wcf.Algorithm(wcf.White(w.ds_target, w.theta_target))) as sm:
        st = sm.iterate(inp.iterations)
    # ... and with the library's default inner tolerance (1e-14: every Picard iterate within 1e-10 RMS of the exact-solve iterate;
    # the reference's own stop test is far looser than either)
    _, mesh2 = _t106_load(None)
    with smooth.Smoother(mesh2, solver.Option.hip(max_inner=40000), wcf.Algorithm(wcf.White(w.ds_target, w.theta_target))) as sm:
        st2 = sm.iterate(inp.iterations)
    return {"gpu_seconds": st["seconds"], "tfi_and_blocking_seconds": t_tfi, "outer_iterations": st["outer_iterations"], "inner_iterations": st["inner_iterations"],
            "not_converged": st["not_converged"], "solver": "hip/bicgstab (diagonal scaling), rtol 1e-8 on the scaled residual, white control function",
            "default_tolerance": {"gpu_seconds": st2["seconds"], "inner_iterations": st2["inner_iterations"], "not_converged": st2["not_converged"], "rtol": 1e-14}}


def solve_to_tolerance(n, smooth, solver, configs, tol=1e-8):
    """BASELINE configs[1] read literally -- "fp64 elliptic smoothing to 1e-8 residual": the perturbed n x n block (SURVEY 8d
    config 2, displacement 0.25 h, seed 12345) driven to a scaled nonlinear residual <= 1e-8 by Picard + multigrid-preconditioned
    BiCGStab.  Reported beside the headline metric, outside its timed region."""
    # inexact Picard: the inner tolerance only has to carry the NONLINEAR residual below `tol` (tools/solve_probe.py: rtol 1e-6
    # reaches it in one outer iteration with 2 inner iterations; 1e-10 needs 5 for the same residual, 1e-4 stalls above it).
    # The job runs twice on identical fresh meshes: the first call of a process also loads ~20 kernel variants (the coarse
    # multigrid levels use their own), ~50 us each on first launch; `seconds` is the second run, the first is reported beside it.
    first = None
    for rep in range(2):
        mesh = configs.single_block(n, n, perturb=0.25)
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-6, check_every=1)) as sm:
            reached, st = sm.iterate_until(tol, 50)
        if rep == 0:
            first = st["seconds"]
    # the criterion in perspective: the UNPERTURBED TFI seed's own scaled residual (rms of the point-Jacobi displacement, ~ h^2)
    seed = configs.single_block(n, n)
    with smooth.Smoother(seed, solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-6, check_every=1)) as sm:
        _, st0 = sm.iterate_until(1e-300, 1)
    return {"reached": bool(reached), "tolerance": tol, "scaled_residual_rms_of_the_unperturbed_tfi_seed": st0["scaled_residual_rms"],
            "outer_iterations": st["outer_iterations"], "inner_iterations": st["inner_iterations"],
            "operator_sweeps": st["operator_sweeps"], "seconds": st["seconds"], "seconds_first_call_in_process": first,
            "scaled_residual_rms": st["scaled_residual_rms"],
            "inner_rtol": 1e-6, "solver": "hip/mg_bicgstab (Picard + BiCGStab, one multigrid V(2,2) cycle per block as preconditioner)"}


def solve_to_converged(n, smooth, solver, configs, tol=1e-10, inexact=False):
    """north_star's "converged node coordinates": the perturbed n x n block, Picard + multigrid-preconditioned BiCGStab with the
    library's default options until the UPDATE of a Picard iteration -- sqrt((sum dx^2 + sum dy^2) / nodes), the quantity the
    reference forms and logs per iteration (smooth.zig:112-137) -- is <= 1e-10.  Reported beside the headline metric, outside its
    timed region; second of two identical runs (the first also loads the kernels)."""
    import numpy as np

    first = None
    for rep in range(2):
        mesh = configs.single_block(n, n, perturb=0.25)
        opt = (solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-1, rtol_initial=True, check_every=1) if inexact
               else solver.Option.hip(inner=solver.Inner.mg_bicgstab))
        with smooth.Smoother(mesh, opt) as sm:
            reached, st = sm.iterate_until_update(tol, 100)
        if rep == 0:
            first = st["seconds"]
    return {"reached": bool(reached), "update_rms_tolerance": tol, "update_rms_last": float(np.sqrt((st["last_dx2"] + st["last_dy2"]) / (n * n))),
            "outer_iterations": st["outer_iterations"], "inner_iterations": st["inner_iterations"], "operator_sweeps": st["operator_sweeps"],
            "seconds": st["seconds"], "seconds_first_call_in_process": first, "scaled_residual_rms_at_last_fill": st["scaled_residual_rms"],
            "criterion": "Picard update rms over all nodes (smooth.zig:112-137) <= 1e-10: distance between consecutive iterates",
            "solver": ("hip/mg_bicgstab, inexact Picard: inner tolerance 0.1 x the initial residual of each solve (TM_OPT_RTOL_INITIAL); same fixed point "
                       "(1e-11 rms from the default-tolerance run, tools/converge_probe.py), the iterates on the way are not the exact-solve ones" if inexact
                       else "hip/mg_bicgstab, default options (inner rtol 7.5e-9 / nodes: every Picard iterate the exact-solve one)")}


KRYLOV_BYTES_PER_NODE = 256.0   # one two-kernel BiCGStab iteration: 176 + 80 B per node


def krylov_iteration(n, smooth, solver, configs, iters=100):
    """The route that actually converges, priced against the same roofline: time per BiCGStab iteration (two kernels, k_apply_vk<VK_R> +
    k_apply_vk<VK_S>, both coordinate components together) of one Picard solve on the perturbed n x n block, capped at `iters` inner
    iterations with no convergence poll in between.  Second of two identical solves (the first loads the kernels)."""
    import torch

    mesh = configs.single_block(n, n, perturb=0.25)
    us = None
    for _ in range(2):
        with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.bicgstab, rtol=1e-30, max_inner=iters, check_every=iters)) as sm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = sm.iterate(1)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) * 1e6 / max(1, st["inner_iterations"])
    gbps = KRYLOV_BYTES_PER_NODE * n * n / (us * 1e-6) / 1e9
    return {"iteration_us": us, "inner_iterations": st["inner_iterations"], "bytes_per_node": KRYLOV_BYTES_PER_NODE, "achieved_GBps": gbps,
            "frac": gbps / HBM_PEAK_GBPS, "what": "hip/bicgstab, two fused kernels per iteration, wall time of one capped Picard solve / iterations (set-up kernels included)"}


def self_launch(args):
    """--gpus N without a launcher: start N fresh ranks as a CHILD process (this process has not touched the GPU and never
    will) and relay rank 0's JSON line."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.lstrip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    raise SystemExit(r.returncode if r.returncode else (0 if lines else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5], help="SURVEY 8d config: 2 = 4096^2 block(s) (default), 4 = 8 x 2048^2 strip strong scaling, 5 = independent 2048^2 slices")
    ap.add_argument("--size", dest="n", type=int, default=0, help="block edge (nodes); default 4096 (config 2) / 2048 (configs 4, 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the (untimed) solve-to-1e-8 and T106 reports")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path (RCCL hooks) even with one rank")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: gather every rank's blocks on rank 0 and compare them, bit for bit, with a single-handle run of the "
                         "whole strip over the same number of sweeps (small sizes; used by the multi-process rehearsals)")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="halo exchange: the library's own RCCL transport (default) or torch.distributed p2p from Python hooks")
    ap.add_argument("--rows", type=int, default=0, help="K2 rows per chunk (tuning)")
    ap.add_argument("--unroll", type=int, default=0, help="K2 row unroll (tuning)")
    ap.add_argument("--pipe", type=int, default=-1, help="K2 software pipelining 0/1 (tuning)")
    ap.add_argument("--nt", type=int, default=-1, help="K2 non-temporal stores 0/1 (tuning)")
    ap.add_argument("--single-sweep", action="store_true", help="one kernel pass per sweep (K2) instead of three (K2x3, fixed walls) or two (K2x2, coupled blocks) sweeps per pass")
    ap.add_argument("--fuse-rows", type=int, default=0, help="K2x2 rows per chunk (tuning)")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed sweeps of the same workload before the W warm-up steps, so that the timed region runs at settled clocks (0 = off)")
    ap.add_argument("--profile-every", type=int, default=PROFILE_EVERY, help="bracket every k-th launch of the dominant kernel with a hipEvent pair")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_dist:
        self_launch(args)   # does not return

    # Libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on stdout, so everything
    # else goes to stderr and the JSON is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    if os.environ.get("TM_BENCH_SAME_DEVICE"):   # rehearsal of the multi-rank path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from turbomesh_amd import _capi, configs
    from turbomesh_amd.smoothing import smooth, solver, wall_control_function as wcf

    if (args.rows or args.unroll or args.pipe >= 0 or args.nt >= 0 or args.fuse_rows) and not hasattr(_capi.lib(), "tm_tune_apply"):
        raise SystemExit("the tuning knobs need the measurement build: TM_HIP_LIB=turbomesh_amd/libtm_hip_dbg.so python bench.py ...")
    if args.rows or args.unroll or args.pipe >= 0 or args.nt >= 0:
        _capi.lib().tm_tune_apply(args.rows, args.unroll, args.pipe, args.nt)
    if args.fuse_rows:
        _capi.lib().tm_tune_fuse(args.fuse_rows)
    relax_opt = solver.Option.hip(inner=solver.Inner.relax, single_sweep=args.single_sweep)

    n = args.n or (4096 if args.config == 2 else 2048)
    scaling = "strong" if args.config == 4 else "weak"
    dist = None
    hooks_obj = None
    owned = [0]
    coupled = args.config != 5 and (world > 1 or args.force_dist)
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod

        dist = dist_mod
        if "MASTER_ADDR" not in os.environ:   # --force-dist without a launcher
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        backend = os.environ.get("TM_BENCH_BACKEND", "nccl")   # "gloo" = rehearsal transport (halo rows staged through the host)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.config == 5:
        per_rank = 8
        nblocks_total = per_rank * world
        mesh = configs.slices(per_rank, n, first=rank * per_rank)          # this rank's slices; nobody else's exist here
        owned = list(range(per_rank))
        sm = smooth.Smoother(mesh, relax_opt, stream=torch.cuda.current_stream().cuda_stream)
        workload = (f"{nblocks_total} independent {n}x{n} slices (SURVEY 8d config 5), {per_rank} per GPU in one handle, replicas only "
                    "(no communication)")
    elif coupled:
        from turbomesh_amd import distributed as tmd

        if args.config == 4:
            nblocks_total = 8
            if nblocks_total % world:
                raise SystemExit("--config 4 shards 8 blocks: --gpus must divide 8")
            bpr = nblocks_total // world
        else:
            nblocks_total, bpr = world, 1
        mesh = tmd.strip_for_rank(world, rank, n, n, blocks_per_rank=bpr)   # only the owned blocks carry coordinates
        owner = [b // bpr for b in range(nblocks_total)]
        owned = [b for b in range(nblocks_total) if owner[b] == rank]
        transport = "torch.distributed p2p (Python hooks)"
        # (TM_RCCL_LIB: the path tm_rccl_* dlopens instead of torch's librccl.  With torch.distributed on gloo -- several ranks on ONE GPU,
        # which real RCCL refuses -- it is what lets the library's own transport run at all: tests/loopback_rccl, test infrastructure.)
        if args.transport == "rccl" and (backend == "nccl" or os.environ.get("TM_RCCL_LIB")):
            # The library's own RCCL transport.  Every rank first checks what it can check ALONE (librccl loads, its symbols
            # resolve, a unique id can be made) and the ranks vote BEFORE any collective call: a rank that failed locally
            # would otherwise leave the others blocked inside ncclCommInitRank.  Then the transport is checked once against
            # the torch.distributed hooks on a small strip (same sweeps, bit-identical coordinates expected).
            ok, why = 1, ""
            try:
                tmd.RcclHooks.precheck()
            except Exception as e:   # noqa: BLE001
                ok, why = 0, repr(e)
            flag_dev = "cuda" if backend == "nccl" else "cpu"
            flag = torch.tensor([ok], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                # the check runs the schedules of the timed region on a small strip: 9 sweeps = three coupled sweep triples (depth-3 halo, one
                # exchange per triple, the fused level kernel), then -- library defaults -- 5 sweeps = a triple and the PAIR that takes the
                # remainder of a sweep count that is not a multiple of three (its exchange, its border pass)
                saved_min = os.environ.get("TM_TRIPLES_MIN_NODES")
                try:
                    # one handle at a time: a process with a single multi-rank handle orders its two queues with counters in
                    # device memory (the schedule of the timed run); two live handles would both fall back to events
                    small = [tmd.strip_for_rank(world, rank, 192, 256, blocks_per_rank=bpr) for _ in range(2)]
                    ok = 1
                    for min_nodes, sweeps in (("1", 9), (None, 5)):
                        if min_nodes is None:
                            os.environ.pop("TM_TRIPLES_MIN_NODES", None)
                            if saved_min is not None:
                                os.environ["TM_TRIPLES_MIN_NODES"] = saved_min
                        else:
                            os.environ["TM_TRIPLES_MIN_NODES"] = min_nodes
                        small = [tmd.strip_for_rank(world, rank, 192, 256, blocks_per_rank=bpr) for _ in range(2)]
                        h_a = tmd.RcclHooks(small[0], owner=owner, rank=rank, world=world, option=relax_opt)
                        h_a.iterate(sweeps)
                        h_a.smoother.download()
                        h_a.close()
                        h_b = tmd.TorchHooks(small[1], owner=owner, rank=rank, world=world, option=relax_opt)
                        h_b.iterate(sweeps)
                        h_b.smoother.download()
                        h_b.smoother.close()
                        ok = ok and int(all(np.array_equal(small[0].blocks[b].points.data, small[1].blocks[b].points.data) for b in owned))
                    why = "" if ok else "coordinates differ from the torch.distributed transport"
                except Exception as e:   # noqa: BLE001 -- any failure means: use the other transport
                    ok, why = 0, repr(e)
                finally:
                    os.environ.pop("TM_TRIPLES_MIN_NODES", None)
                    if saved_min is not None:
                        os.environ["TM_TRIPLES_MIN_NODES"] = saved_min
                flag = torch.tensor([ok], dtype=torch.int32, device=flag_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_native = int(flag.item()) == 1
            if use_native:
                transport = "RCCL p2p issued by libtm_hip (tm_rccl_*)" + (f" through $TM_RCCL_LIB = {os.path.basename(os.environ['TM_RCCL_LIB'])}" if os.environ.get("TM_RCCL_LIB") else "")
            else:
                print(f"[bench] rank {rank}: native RCCL transport not used ({why or 'another rank declined'})", file=sys.stderr)
        else:
            use_native = False

        def make_coupled_handle():
            if use_native:
                return tmd.RcclHooks(mesh, owner=owner, rank=rank, world=world, option=relax_opt)
            return tmd.TorchHooks(mesh, owner=owner, rank=rank, world=world, option=relax_opt)

        hooks_obj = make_coupled_handle()
        sm = hooks_obj.smoother
        workload = (f"strip of {nblocks_total} coupled blocks {n}x{n} (SURVEY 8d config {args.config}), {bpr} per GPU, interface rows exchanged by "
                    f"{transport} between sweeps")
    elif args.config == 4:
        nblocks_total = 8
        mesh = configs.strip(nblocks_total, n, n)
        owned = list(range(nblocks_total))
        sm = smooth.Smoother(mesh, relax_opt, stream=torch.cuda.current_stream().cuda_stream)
        workload = f"strip of 8 coupled blocks {n}x{n} (SURVEY 8d config 4), all on one GPU"
    else:
        nblocks_total = 1
        mesh = configs.single_block(n, n)                                   # TFI on the GPU (K1)
        sm = smooth.Smoother(mesh, relax_opt, stream=torch.cuda.current_stream().cuda_stream)
        workload = f"single synthetic {n}x{n} block, TFI seed, Laplace control function, fixed boundary (SURVEY 8d config 2)"

    nodes_rank_est = n * n * len(owned)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up.  A multi-rank sweep pair orders its two queues with counters in device memory; a wait that is not
    # met within its limit fails the pass with TM_E_HIP instead of hanging the device.  That has never been observed, but the schedule
    # has never met real RCCL kernels from several peers on its queue either: if it happens on ANY rank, all ranks re-create their
    # handles with the event-ordered schedule (TM_PAIR_SYNC=events, slower by ~10-25 % per pair) and the line says so.
    pair_sync = os.environ.get("TM_PAIR_SYNC", "counters") if coupled else None
    # Settling.  The sweep runs at the board's power cap (1380 W of 1400 W, shader clock 2.0 of 2.4 GHz: tools/dev/clock_probe.py), and
    # the power controller needs ~50 ms of load before the clock has settled (three sweeps per pass: 54.7 us per sweep in the first
    # 10 ms, 43.5 from 50 ms on, tools/dev/ramp_probe.py) -- far longer than W + K steps of the driver's runs.  The SAME sweeps on the
    # SAME handle therefore run untimed for ~0.25 s first (a count fixed by the workload's size, identical on every rank); W and
    # K are what the flags say and the line reports the extra steps.  --settle-ms 0 turns it off.
    est_us = 55.0 * nodes_rank_est / (4096.0 * 4096.0)
    settle_steps = 0 if args.settle_ms <= 0 else max(60, min(60000, int(args.settle_ms * 1e3 / max(est_us, 0.5)) // 6 * 6))
    ok = 1
    cold_dt = None
    pre_steps = (args.warmup + args.steps + settle_steps) if settle_steps else 0   # untimed sweeps in front of the W warm-up steps
    try:
        if settle_steps:
            # the figure a W + K run WITHOUT the settling phase reads (kernels loaded by the W steps, clocks not yet settled):
            # reported as config.cold_ms_per_step beside ms_per_step, never used for `value`
            sm.iterate(args.warmup)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            sm.iterate(args.steps)
            torch.cuda.synchronize()
            cold_dt = time.perf_counter() - tc
            sm.iterate(settle_steps)
        sm.iterate(args.warmup)
        torch.cuda.synchronize()
    except _capi.TmError as e:
        if not (coupled and e.code == _capi.TM_E_HIP and pair_sync != "events"):
            raise
        print(f"[bench] rank {rank}: {e}", file=sys.stderr)
        ok = 0
    if coupled and dist is not None:
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        if hasattr(hooks_obj, "close"):
            hooks_obj.close()
        else:
            hooks_obj.smoother.close()
        os.environ["TM_PAIR_SYNC"] = "events"      # read when a handle is created
        mesh = tmd.strip_for_rank(world, rank, n, n, blocks_per_rank=bpr)
        hooks_obj = make_coupled_handle()
        sm = hooks_obj.smoother
        sm.iterate(settle_steps + args.warmup)
        pre_steps, cold_dt = settle_steps, None
        pair_sync = "events (fallback: a device-side wait of the counter-ordered schedule timed out during warm-up)"
    sm.profile(max(1, args.profile_every))
    barrier()
    t0 = time.perf_counter()
    st = sm.iterate(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    k2_ms, k2_timed, k2_launches = sm.profile_read()
    sm.profile(0)

    verified = None
    if args.verify and dist is not None and coupled:
        sm.download()
        mine = torch.from_numpy(np.stack([mesh.blocks[b].points.data for b in owned]).copy())
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        if dist.get_backend() == "nccl":
            mine = mine.cuda()
            parts = [p.cuda() for p in parts] if parts else None
        dist.gather(mine, parts, dst=0)
        if rank == 0:
            whole = configs.strip(nblocks_total, n, n)
            with smooth.Smoother(whole, relax_opt) as ref:
                if pre_steps:
                    ref.iterate(pre_steps)
                ref.iterate(args.warmup)
                ref.iterate(args.steps)
                ref.download()
            got = np.concatenate([p.cpu().numpy() for p in parts])
            verified = all(np.array_equal(got[b], whole.blocks[b].points.data) for b in range(nblocks_total))
            print(f"[bench] --verify: {world} ranks vs one handle after {pre_steps}+{args.warmup}+{args.steps} sweeps: {'bit-identical' if verified else 'MISMATCH'}", file=sys.stderr)

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    nodes_total = n * n * nblocks_total
    nodes_rank = n * n * len(owned)
    value = nodes_total * args.steps / dt

    if rank == 0:
        k2_avg_s = (k2_ms / 1e3) / max(1, k2_timed)
        sweeps_per_launch = st["operator_sweeps"] / max(1, k2_launches)   # one launch takes this rank's blocks through 1 (K2), 2 (K2x2) or 3 (K2x3) sweeps
        whole_job_timing = coupled or args.config == 4
        if whole_job_timing:
            # coupled blocks: the interior part of a pass waits in its queue for the border part of the previous one, so an event pair
            # around it times the schedule, not the kernel -- use the whole job's time per pass instead (perimeter-row kernels included)
            k2_avg_s = dt / args.steps * sweeps_per_launch
        fused = sweeps_per_launch > 1.5
        spl = int(round(sweeps_per_launch))
        kname = {1: "k_apply", 2: "k_relax2", 3: "k_relax3"}.get(spl, "k_relax2")
        # what ONE launch of the dominant kernel has to move: every owned node read once and written once, however many
        # sweeps it performs on the way (K2x2: two, K2x3: three).  achieved / peak is therefore a true bandwidth fraction, <= 1.
        bytes_per_launch = BYTES_PER_NODE * nodes_rank
        achieved = bytes_per_launch / k2_avg_s / 1e9
        traffic, traffic_src, valu_insts = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # written by tools/pmc_traffic.py from separate rocprofv3 --pmc passes
        if os.path.exists(tpath) and args.config == 2 and world == 1:
            try:
                tj = json.load(open(tpath))
                if tj.get("n") == n and tj.get("kernel", "k_apply") == kname:
                    stale = tj.get("kernels_hash") not in (None, kernels_hash())
                    traffic = None if stale else tj.get("hbm_bytes_per_launch")
                    valu_insts = None if stale else tj.get("valu_wave_insts_per_launch")
                    traffic_src = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of an earlier run of this command"
                                   + (f", {tj.get('date')}" if tj.get("date") else "") + ("; STALE (kernel source changed since), dropped" if stale else ""))
            except Exception:
                traffic = None
        # the measured ceiling beside the specification: copy / triad at this workload's footprint, in this run
        import ctypes as C

        cp, tr = C.c_double(0), C.c_double(0)
        try:
            _capi.check(_capi.lib().tm_stream_probe(int(16 * nodes_rank), 20, C.byref(cp), C.byref(tr)))
            stream = {"copy": cp.value, "triad": tr.value}
        except Exception as e:   # noqa: BLE001 -- a diagnostic must not cost the line
            print(f"[bench] stream probe failed: {e}", file=sys.stderr)
            stream = None
        # north_star's "single 4096^2 elliptic sweep >= 60 % of the HBM roofline" in the same line: ONE sweep per launch (K2) on the same
        # block, 96 launches event-timed outside the timed region, behind 480 untimed ones (that pass is bandwidth-bound and runs below the power cap)
        single_ref = None
        if world == 1 and args.config == 2 and fused:
            try:
                m1 = configs.single_block(n, n)
                with smooth.Smoother(m1, solver.Option.hip(inner=solver.Inner.relax, single_sweep=True), stream=torch.cuda.current_stream().cuda_stream) as s1:
                    s1.iterate(480)   # its own settling: a fresh handle, and this pass draws less power than the one just timed
                    s1.profile(max(1, args.profile_every))
                    s1.iterate(96)
                    ms1, timed1, _ = s1.profile_read()
                us1 = ms1 * 1e3 / max(1, timed1)
                single_ref = {"kernel": "k_apply<RELAX,DELTA,field,laplace> (K2: one sweep per launch)", "avg_launch_us": us1,
                              "achieved_GBps": BYTES_PER_NODE * n * n / (us1 * 1e-6) / 1e9, "frac": BYTES_PER_NODE * n * n / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            except Exception as e:   # noqa: BLE001
                print(f"[bench] single-sweep reference failed: {e}", file=sys.stderr)
        # The pass is bound by fp64 ISSUE, not by bytes: price the same launch time against the vector unit as well.  A wave64 fp64 (or
        # DPP) instruction occupies a SIMD for 4 cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 4 = 614.4 G wave-instructions/s -- the figure behind
        # the part's 78.6 TFLOP/s fp64 vector peak (MI355X_MICROARCH.md).  Instructions per launch: SQ_INSTS_VALU of a separate --pmc pass.
        valu = None
        if valu_insts:
            peak_ginst = 256 * 4 * 2.4 / 4.0
            ach = valu_insts / k2_avg_s / 1e9
            valu = {"wave_instructions_per_launch": valu_insts, "achieved_Ginst_per_s": ach, "peak_Ginst_per_s": peak_ginst, "frac": ach / peak_ginst,
                    "per_node_and_sweep": valu_insts * 64.0 / (nodes_rank * spl),
                    "what": ("VALU wave-instructions (SQ_INSTS_VALU, profiles/traffic.json, same kernel source) / this run's launch time, against the issue peak at the "
                             "2.4 GHz specification clock; the pass runs at ~2.0 GHz under the 1400 W cap, where the same count is ~0.8 of what can issue; "
                             "the reciprocal (v_rcp_f64) issues at a quarter of that rate, so the pipes are busier than the instruction count says")}
        out = {
            "metric": f"nodes smoothed/sec (elliptic sweeps, {n}^2 blocks) + achieved HBM GB/s",
            "value": value, "unit": "nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "nodes_total": nodes_total, "nodes_per_gpu": nodes_rank,
                       "solver": "hip/relax (fused Jacobi elliptic sweep" + (f", {spl} sweeps per kernel pass)" if fused else ")"), "omega": 1.0,
                       "settle_steps": settle_steps, "cold_ms_per_step": (cold_dt / args.steps * 1e3 if cold_dt else None),
                       "cold_value": (nodes_rank * args.steps / cold_dt if cold_dt and dist is None else None),
                       "settle": {"steps": settle_steps, "why": "untimed sweeps of the same workload before the W warm-up steps: the pass runs at the board power cap and "
                                  "the clock needs ~50 ms of load to settle (tools/dev/ramp_probe.py); --settle-ms 0 turns it off"},
                       "residual_last": st["last_residual"], **({"verified_against_single_handle": verified} if verified is not None else {}),
                       **({"pair_sync": pair_sync, "queue_ordering": sm.queue_ordering()[1]} if pair_sync else {}),
                       "sweep_equiv_GBps_whole_job": BYTES_PER_NODE * value / 1e9},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "valu_issue": valu,
                         "valu_issue_frac": (valu["frac"] if valu else None),
                         "single_sweep_reference": single_ref,
                         "single_sweep_us": (single_ref["avg_launch_us"] if single_ref else None),
                         "single_sweep_frac": (single_ref["frac"] if single_ref else None),
                         "stream_copy_GBps": (stream["copy"] if stream else None), "stream_triad_GBps": (stream["triad"] if stream else None),
                         "stream_ceiling_GBps": (max(stream.values()) if stream else None),
                         "frac_of_stream": (achieved / max(stream.values()) if stream else None),
                         "stream": ({**stream, "what": f"tm_stream_probe in this run: copy / triad of {16 * nodes_rank / 2**20:.0f} MiB per array, non-temporal 16 B/lane, 20 launches each"}
                                    if stream else None),
                         "kernel": (f"{kname}<DELTA> (K2x{spl}: {spl} winslow sweeps per launch)" if fused else "k_apply<RELAX,DELTA,field,laplace> (K2 winslow_apply)"),
                         "limited_by": ("fp64 issue under the board power cap (1380 W of 1400 W, shader clock 2.0 of 2.4 GHz while the pass runs: tools/dev/clock_probe.py); "
                                        "more sweeps per pass trade HBM bytes for nothing but the arithmetic the sweeps need anyway, so `frac` falls while nodes/s rise"
                                        if fused else "HBM"),
                         "sweeps_per_launch": sweeps_per_launch, "bytes_per_launch": bytes_per_launch,
                         "bytes_model": ("32 B per owned node per LAUNCH: the field is read once and written once per pass (rank 0's blocks); the residual-norm "
                                         "partial sums ride only in the LAST launch of an iterate() call (the one whose norms are read back; measured cost nil: "
                                         "k_relax3<DOT_DELTA> 115.1 us against 115.06 without)"),
                         "sweep_equiv_GBps": BYTES_PER_NODE * nodes_rank * sweeps_per_launch / k2_avg_s / 1e9,
                         "sweep_equiv_note": ("SURVEY 8d counts 32 B per node per SWEEP; a temporally blocked pass performs several sweeps for one read + one write, "
                                              "so this figure may exceed the HBM peak -- it is not a bandwidth") if fused else "one sweep per launch: equals achieved",
                         "hbm_GBps_measured_traffic": (traffic / k2_avg_s / 1e9) if traffic else None,
                         "avg_launch_us": k2_avg_s * 1e6, "launches": k2_launches, "launches_timed": k2_timed,
                         "timing": ("coupled blocks: whole-job time per pass (timed region / passes), perimeter-row kernels and the schedule's gaps included" if whole_job_timing else
                                    f"hipEvent pairs on the handle's stream inside the timed region: around alternate groups of {max(1, args.profile_every)} consecutive "
                                    "launches of the kernel where nothing else is launched between them (a block with fixed walls: an event record is a "
                                    f"barrier packet of several us), else around every {max(1, args.profile_every)}th launch")},
        }
        if scaling == "strong":
            ref1 = None
            p1 = os.path.join(ROOT, "profiles", "config4_n1.json")
            if os.path.exists(p1):
                try:
                    ref1 = json.load(open(p1)).get("value")
                except Exception:
                    ref1 = None
            out["config"]["value_1gpu_stored"] = ref1
            out["config"]["vs_1gpu"] = (value / ref1) if ref1 else None
        if world == 1 and args.config == 2 and not args.no_solve:
            c = out["config"]
            c["solve_to_1e-8"] = solve_to_tolerance(n, smooth, solver, configs)
            c["solve_to_converged"] = solve_to_converged(n, smooth, solver, configs)
            c["solve_to_converged_inexact_picard"] = solve_to_converged(n, smooth, solver, configs, inexact=True)
            # the same figures as scalars (a reader that keeps only two levels of the line still sees them)
            c["solve_to_1e-8_ms"] = c["solve_to_1e-8"]["seconds"] * 1e3
            c["solve_to_converged_ms"] = c["solve_to_converged"]["seconds"] * 1e3
            c["solve_to_converged_outer"] = c["solve_to_converged"]["outer_iterations"]
            c["solve_to_converged_inner"] = c["solve_to_converged"]["inner_iterations"]
            c["solve_to_converged_inexact_ms"] = c["solve_to_converged_inexact_picard"]["seconds"] * 1e3
            try:
                kr = krylov_iteration(n, smooth, solver, configs)
                out["roofline"]["krylov"] = kr
                out["roofline"]["krylov_iter_us"] = kr["iteration_us"]
                out["roofline"]["krylov_frac"] = kr["frac"]
            except Exception as e:   # noqa: BLE001 -- a diagnostic must not cost the line
                print(f"[bench] Krylov iteration probe failed: {e}", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1 and args.config == 2:
            cb = cpu_baseline(n)
            if not args.no_solve and "t106_json_as_written" in cb:
                cb["t106_json_as_written"]["gpu_same_job"] = t106_gpu(smooth, solver, wcf)
                cb["t106_cpu_seconds"] = cb["t106_json_as_written"]["cpu_seconds"]
                cb["t106_gpu_seconds"] = cb["t106_json_as_written"]["gpu_same_job"]["gpu_seconds"]
            cb["mirror_sweep_nodes_per_s_1_thread"] = cb["mirror_sweep"]["value"]
            cb["mirror_sweep_nodes_per_s_all_threads"] = cb["mirror_sweep"]["all_threads"]["value"]
            out["cpu_baseline"] = cb
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if hooks_obj is not None:   # communicators go before the process group (their teardown is collective)
        if hasattr(hooks_obj, "close"):
            hooks_obj.close()
        else:
            hooks_obj.smoother.close()
    elif sm is not None:
        sm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
