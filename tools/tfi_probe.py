#!/usr/bin/env python3
"""K1 (TFI) on device pointers: time per 4096^2 block (write-only, 16 B/node)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from turbomesh_amd import _capi, configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = configs.single_block_edges(n, n)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pts = [dev(x.points) for x in e]; cl = [dev(x.clustering) for x in e]
out = torch.empty(n * n * 2, dtype=torch.float64, device="cuda")
L = _capi.lib()
p = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
def run():
    _capi.check(L.tm_dev_tfi_block(p(out), n, n, p(pts[0]), p(pts[1]), p(pts[2]), p(pts[3]), p(cl[0]), p(cl[1]), p(cl[2]), p(cl[3]), None))
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"TFI {n}^2: {us:.1f} us per block = {16.0 * n * n / us / 1e3:.0f} GB/s written")
