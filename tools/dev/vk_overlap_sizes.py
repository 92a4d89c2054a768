#!/usr/bin/env python3
"""us per BiCGStab iteration by block size with and without the overlapping-strip layout of the two Krylov kernels (TM_VK_OVERLAP), alternating
in one process-pair per size so that both figures come from the same box and the same minute.   usage: vk_overlap_sizes.py [sizes...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import os, sys, time
import torch
sys.path.insert(0, os.environ['TM_ROOT'])
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
n = int(sys.argv[1]); iters = max(100, min(600, int(3e9 / (n * n * 16))))
mesh = configs.single_block(n, n, perturb=0.25)
best = 1e30
for rep in range(3):
    with smooth.Smoother(mesh, solver.Option.hip(rtol=1e-30, max_inner=iters, check_every=iters)) as sm:
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = sm.iterate(1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rep: best = min(best, dt / st["inner_iterations"] * 1e6)
print(best)
"""
sizes = [int(a) for a in sys.argv[1:]] or [1024, 1448, 2048, 2896, 4096, 5792]
for n in sizes:
    row = []
    for rep in range(2):
        for ov in ("0", "1"):
            r = subprocess.run([sys.executable, "-c", CHILD, str(n)], capture_output=True, text=True, env=dict(os.environ, TM_VK_OVERLAP=ov, TM_ROOT=ROOT))
            row.append(float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else float("nan"))
    print(f"{n:5d}^2   halo loads {row[0]:8.1f} {row[2]:8.1f}   overlapping strips {row[1]:8.1f} {row[3]:8.1f}   ratio {min(row[1], row[3]) / min(row[0], row[2]):.3f}", flush=True)
