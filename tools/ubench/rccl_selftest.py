#!/usr/bin/env python3
"""Device-side and host-side cost of the library's grouped ncclSend/ncclRecv exchange, with the own rank as the only peer."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("TM_HIP_LIB", os.path.join(sys.path[0], "turbomesh_amd", "libtm_hip_dbg.so"))   # measurement build: tm_debug_* / tm_tune_* / tm_diag_*
import torch
from turbomesh_amd import _capi
L = _capi.lib()
L.tm_debug_rccl_selftest.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
torch.cuda.set_device(0)
path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
uid = (C.c_ubyte * 128)()
_capi.check(L.tm_rccl_unique_id(path, uid))
comm = C.c_void_p()
_capi.check(L.tm_rccl_comm_create(path, uid, 0, 1, C.byref(comm)))
for rows in (2048, 4096, 16384):
    us = C.c_double(0)
    t0 = time.perf_counter()
    _capi.check(L.tm_debug_rccl_selftest(comm, rows, 300, C.byref(us)))
    wall = (time.perf_counter() - t0) / 310 * 1e6
    print(f"{rows} rows ({rows * 16 // 1024} KiB per message, 2 sends + 2 recvs): {us.value:.1f} us per exchange on the device, {wall:.1f} us wall per call", flush=True)
L.tm_rccl_comm_destroy(comm)
