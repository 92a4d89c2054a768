#!/usr/bin/env python3
"""Shader clock and power while K2x2 runs flat out on a 4096^2 block (is the pass power-limited?): samples rocm-smi from a thread.
usage: clock_probe.py [seconds = 8] [single | bicgstab | mg]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
mode = sys.argv[2] if len(sys.argv) > 2 else "relax"
single = mode == "single"
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "fclk", "junction"))]
            samples.append((time.time(), keep))
        except Exception as e:
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.5)
mesh = configs.single_block(4096, 4096, perturb=0.25 if mode in ("bicgstab", "mg") else 0.0)
if mode in ("bicgstab", "mg"):   # Krylov kernels: Picard solves capped at 400 inner iterations, tolerance out of reach
    opt = solver.Option.hip(inner=solver.Inner.mg_bicgstab if mode == "mg" else solver.Inner.bicgstab, rtol=1e-30, max_inner=400 if mode == "bicgstab" else 40, check_every=400)
    with smooth.Smoother(mesh, opt) as sm:
        sm.iterate(1)
        th = threading.Thread(target=sampler); th.start()
        t0 = time.time(); per = []
        while time.time() - t0 < secs:
            st = sm.iterate(1)
            per.append(st["seconds"] / max(1, st["inner_iterations"]) * 1e6)
        stop = True; th.join()
    print("us per inner iteration over time:", " ".join(f"{p:.0f}" for p in per))
else:
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax, single_sweep=single)) as sm:
        sm.iterate(100)
        th = threading.Thread(target=sampler); th.start()
        t0 = time.time(); per = []
        while time.time() - t0 < secs:
            st = sm.iterate(4000)
            per.append(st["seconds"] / 4000 * 1e6)
        stop = True; th.join()
    print("us per sweep over time:", " ".join(f"{p:.1f}" for p in per))
t00 = samples[0][0] if samples else 0
for t, keep in samples:
    print(f"t+{t - t00:4.1f}s", " | ".join(keep))
