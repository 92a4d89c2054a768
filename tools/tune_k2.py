#!/usr/bin/env python3
"""Sweep the K2 launch parameters (rows per chunk, row unroll) on the 4096^2 relax sweep and print the
average kernel time from hipEvent pairs; also times a plain device copy of the same footprint
(the achievable HBM ceiling on this box)."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TM_HIP_LIB", os.path.join(sys.path[0], "turbomesh_amd", "libtm_hip_dbg.so"))   # measurement build: tm_debug_* / tm_tune_* / tm_diag_*
import torch

from turbomesh_amd import _capi, configs
from turbomesh_amd.smoothing import smooth, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = 60
res = {}
# copy ceiling: read 256 MiB + write 256 MiB
a = torch.empty(n * n * 2, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
for _ in range(5):
    b.copy_(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    b.copy_(a)
e1.record()
torch.cuda.synchronize()
copy_us = e0.elapsed_time(e1) * 1e3 / 50
res["copy"] = {"us": copy_us, "GBps": 32.0 * n * n / copy_us / 1e3}
print("copy", res["copy"], flush=True)
del a, b

# ---- diagnostics: K2 tiling with reduced arithmetic, and the stand-alone relax sweep, timed with torch events
def timed(fn, reps=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

a = torch.empty(n * n * 2, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
L = _capi.lib()
for rows, unroll, pipe, nt in ((64, 3, 0, 1), (32, 6, 0, 1)):
    L.tm_tune_apply(rows, unroll, pipe, nt)
    for mode, label in ((4, "diag_copy"), (5, "diag_sum9"), (6, "diag_nostore(math+loads)"), (7, "diag_noload(math+stores)"), (8, "diag_math_only")):
        us = timed(lambda: L.tm_diag_apply(a.data_ptr(), b.data_ptr(), n, n, mode, None))
        print(os.environ.get("TM_TUNE_TAG", ""), label, "rows", rows, "U", unroll, "pipe", pipe, "nt", nt, f"{us:.1f} us (back-to-back launches incl. gaps)", f"{32.0 * n * n / us / 1e3:.0f} GB/s", flush=True)
    us = timed(lambda: L.tm_dev_relax_sweep(a.data_ptr(), b.data_ptr(), n, n, 1.0, None, 0, None, None))
    print(os.environ.get("TM_TUNE_TAG", ""), "dev_relax_sweep(+perimeter copy)", rows, unroll, pipe, nt, f"{us:.1f} us", flush=True)
del a, b

mesh = configs.single_block(n, n)
tag = os.environ.get("TM_TUNE_TAG", "default")
# Krylov apply (48-64 B/node: frozen field + vector in, vector out) through the BiCGStab path
for rows, unroll, pipe in ():
    _capi.lib().tm_tune_apply(rows, unroll, pipe, 0)
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.bicgstab, max_inner=24, rtol=1e-30)) as sm:
        sm.profile(True)
        st = sm.iterate(1)
        ms, k, _ = sm.profile_read()
    us = ms * 1e3 / k
    res[f"krylov_r{rows}_u{unroll}_p{pipe}"] = {"us": us, "launches": k, "iter_seconds": st["seconds"], "inner": st["inner_iterations"]}
    print(tag, "krylov apply", rows, unroll, pipe, f"{us:.1f} us avg over {k} launches; 1 Picard with {st['inner_iterations']} inner its took {st['seconds']*1e3:.1f} ms", flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"tune_k2_{tag}.json"), "w"), indent=1)
