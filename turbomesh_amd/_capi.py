"""ctypes binding of libtm_hip.so (include/tm_hip.h) -- the drop-in C-ABI boundary.

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device
is present, calls raise TmError / OSError.  Only `plan_build` (host-side planning) works
without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TM_HIP_LIB") or os.path.join(_HERE, "libtm_hip.so")   # TM_HIP_LIB: experiment builds only
DBG_LIB_PATH = os.path.join(_HERE, "libtm_hip_dbg.so")   # the same library + the tm_debug_* / tm_tune_* / tm_diag_* helpers of tools/

# tm_error
TM_OK, TM_W_NOT_CONVERGED = 0, 1
TM_E_SIZE, TM_E_TOPOLOGY, TM_E_MISMATCH, TM_E_OVERFLOW, TM_E_UNSUPPORTED, TM_E_ARG, TM_E_MEMORY, TM_E_HIP, TM_E_COMM = -1, -2, -3, -4, -5, -6, -7, -8, -9
_ERR_NAMES = {-1: "InconsistentSize", -2: "Topology", -3: "Mismatch", -4: "Overflow", -5: "ExternalSolverNotEnabled", -6: "Argument",
              -7: "OutOfMemory", -8: "Hip", -9: "Comm"}
TM_SOLVER_GMRES, TM_SOLVER_BICGSTAB, TM_SOLVER_UMFPACK, TM_SOLVER_PETSC, TM_SOLVER_HIP = 0, 1, 2, 3, 4
TM_INNER_BICGSTAB, TM_INNER_RELAX, TM_INNER_MG_BICGSTAB, TM_INNER_AUTO, TM_INNER_GMRES = 0, 1, 2, 3, 4
TM_CF_LAPLACE, TM_CF_WHITE = 0, 1


class TmError(RuntimeError):
    """A negative tm_error code mapped to a Python error (the Zig side maps it to an error union)."""

    def __init__(self, code, msg):
        super().__init__(f"error.{_ERR_NAMES.get(code, 'Unknown')} ({code}): {msg}")
        self.code = code
        self.name = _ERR_NAMES.get(code, "Unknown")


class tm_range(C.Structure):
    _fields_ = [("block", C.c_uint64), ("side", C.c_uint32), ("_pad", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64)]


class tm_connection(C.Structure):
    _fields_ = [("r", tm_range * 2), ("has_periodicity", C.c_int32), ("_pad", C.c_int32), ("periodicity", C.c_double * 2)]


class tm_condition(C.Structure):
    _fields_ = [("range", tm_range), ("kind", C.c_uint32), ("_pad", C.c_uint32)]


class tm_block(C.Structure):
    _fields_ = [("xy", C.POINTER(C.c_double)), ("ni", C.c_uint64), ("nj", C.c_uint64)]


class tm_mesh_desc(C.Structure):
    _fields_ = [("blocks", C.POINTER(tm_block)), ("nblocks", C.c_uint64), ("conns", C.POINTER(tm_connection)), ("nconns", C.c_uint64),
                ("bcs", C.POINTER(tm_condition)), ("nbcs", C.c_uint64)]


class tm_control_fn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("ds_target", C.c_double), ("theta_target", C.c_double)]


class tm_solver_opt(C.Structure):
    _fields_ = [("tag", C.c_int32), ("inner", C.c_int32), ("rtol", C.c_double), ("atol", C.c_double), ("max_inner", C.c_uint64),
                ("check_every", C.c_uint32), ("flags", C.c_uint32), ("omega", C.c_double)]


class tm_stats(C.Structure):
    _fields_ = [("outer_iterations", C.c_uint64), ("inner_iterations", C.c_uint64), ("operator_sweeps", C.c_uint64),
                ("last_residual", C.c_double), ("last_dx2", C.c_double), ("last_dy2", C.c_double), ("scaled_residual_rms", C.c_double),
                ("seconds", C.c_double), ("not_converged", C.c_int32), ("_pad", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("_")}


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
EXCHANGE_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_uint64, C.c_double)


class tm_comm_hooks(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("nranks", C.c_int32), ("owner", C.POINTER(C.c_int32)),
                ("exchange", EXCHANGE_FN), ("allreduce_sum", ALLREDUCE_FN), ("exchange_wait", EXCHANGE_WAIT_FN), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_uint64)]


class tm_plan_rows(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("row", C.POINTER(C.c_int64)), ("kind", C.POINTER(C.c_int32)), ("ncols", C.POINTER(C.c_int32)),
                ("cols", C.POINTER(C.c_int64)), ("coef_x", C.POINTER(C.c_double)), ("coef_y", C.POINTER(C.c_double)),
                ("rhs", C.POINTER(C.c_double)), ("slot", C.POINTER(C.c_int32))]


class tm_plan_local_info(C.Structure):
    _fields_ = [("n_owned", C.c_int64), ("n_ghost", C.c_int64), ("n_send", C.c_int64), ("npeers", C.c_int32), ("nowned_blocks", C.c_int32),
                ("owned_blocks", C.POINTER(C.c_int64)), ("local_start", C.POINTER(C.c_int64)), ("ghost_gid", C.POINTER(C.c_int64)),
                ("send_ids", C.POINTER(C.c_int32)), ("send_gid", C.POINTER(C.c_int64)), ("peer_rank", C.POINTER(C.c_int32)),
                ("send_offset", C.POINTER(C.c_int64)), ("send_count", C.POINTER(C.c_int64)), ("recv_offset", C.POINTER(C.c_int64)),
                ("recv_count", C.POINTER(C.c_int64)), ("send_first", C.POINTER(C.c_int64)), ("direct_send", C.c_int32), ("_pad", C.c_int32),
                ("n_ghost_rows", C.c_int64), ("ghost_row_gid", C.POINTER(C.c_int64)), ("ghost_row_kind", C.POINTER(C.c_int32)),
                ("ghost_row_cols", C.POINTER(C.c_int64))]


class tm_rccl_peer_table(C.Structure):
    _fields_ = [("npeers", C.c_int32), ("direct_send", C.c_int32), ("send_rows", C.c_int64), ("recv_rows", C.c_int64),
                ("peer", C.POINTER(C.c_int32)), ("send_off", C.POINTER(C.c_int64)), ("send_cnt", C.POINTER(C.c_int64)),
                ("recv_off", C.POINTER(C.c_int64)), ("recv_cnt", C.POINTER(C.c_int64))]


# every symbol include/tm_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "tm_last_error", "tm_abi_version", "tm_set_log", "tm_csr_solve", "tm_tfi_block", "tm_tfi_linear2d", "tm_smooth_mesh", "tm_smoother_create",
    "tm_smoother_workspace_bytes", "tm_smoother_iterate", "tm_smoother_iterate_until", "tm_smoother_iterate_until_update", "tm_smoother_download", "tm_smoother_upload", "tm_smoother_destroy",
    "tm_smoother_exchange_plan", "tm_smoother_apply", "tm_smoother_rhs", "tm_smoother_row_kinds", "tm_smoother_dof",
    "tm_smoother_control_function", "tm_smoother_profile", "tm_smoother_profile_read", "tm_plan_build", "tm_plan_free", "tm_plan_local", "tm_plan_local_free", "tm_dev_tfi_block", "tm_dev_relax_sweep",
    "tm_dev_relax_partials_needed", "tm_export_soa", "tm_smoother_export_soa", "tm_rccl_unique_id", "tm_rccl_comm_create", "tm_rccl_comm_destroy", "tm_rccl_hooks",
    "tm_rccl_peer_table_build", "tm_rccl_peer_table_free", "tm_white_math_probe", "tm_stream_probe", "tm_smoother_queue_ordering", "tm_smoother_inner", "tm_csr_ilu0_probe", "tm_rccl_hooks_for", "tm_smoother_assemble_csr", "tm_smoother_apply_reference_order",
]

_lib = None
_dp = C.POINTER(C.c_double)


def _share_hip_runtime():
    """A process must run on ONE HIP runtime.  libtm_hip.so is linked against the system's libamdhip64; PyTorch ships its own
    copy and, loaded second, finds no GPU ("No HIP GPUs are available").  When torch is installed but not imported yet, its
    copy is loaded first so that both resolve to it -- the order every test and bench.py already has."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        try:
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libtm_hip.so; raises OSError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C turbomesh_amd/csrc`; turbomesh_amd has no CPU fallback")
        _share_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.tm_last_error.restype = C.c_char_p
        L.tm_csr_solve.argtypes = [C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(tm_solver_opt),
                                   C.POINTER(tm_stats)]
        L.tm_set_log.argtypes = [LOG_FN, C.c_void_p]
        L.tm_set_log.restype = None
        L.tm_smoother_dof.restype = C.c_uint64
        L.tm_smoother_dof.argtypes = [C.c_void_p]
        L.tm_dev_relax_partials_needed.restype = C.c_uint64
        L.tm_dev_relax_partials_needed.argtypes = [C.c_uint64, C.c_uint64]
        L.tm_tfi_block.argtypes = [_dp, C.c_uint64, C.c_uint64] + [_dp] * 8
        L.tm_tfi_linear2d.argtypes = [_dp, C.c_uint64, C.c_uint64] + [_dp] * 4
        L.tm_smooth_mesh.argtypes = [C.POINTER(tm_mesh_desc), C.c_uint64, C.POINTER(tm_solver_opt), C.POINTER(tm_control_fn), C.POINTER(tm_stats)]
        L.tm_smoother_create.argtypes = [C.POINTER(tm_mesh_desc), C.POINTER(tm_solver_opt), C.POINTER(tm_control_fn), C.POINTER(tm_comm_hooks),
                                         C.c_void_p, C.POINTER(C.c_void_p)]
        L.tm_smoother_workspace_bytes.argtypes = [C.POINTER(tm_mesh_desc), C.POINTER(tm_solver_opt), C.POINTER(tm_control_fn),
                                                  C.POINTER(tm_comm_hooks), C.POINTER(C.c_uint64)]
        L.tm_smoother_iterate.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(tm_stats)]
        L.tm_smoother_iterate_until.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.POINTER(tm_stats)]
        L.tm_smoother_iterate_until_update.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.POINTER(tm_stats)]
        L.tm_smoother_download.argtypes = [C.c_void_p, C.POINTER(tm_mesh_desc)]
        L.tm_smoother_upload.argtypes = [C.c_void_p, C.POINTER(tm_mesh_desc)]
        L.tm_smoother_destroy.argtypes = [C.c_void_p]
        L.tm_smoother_destroy.restype = None
        L.tm_smoother_exchange_plan.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32))] + [C.POINTER(C.POINTER(C.c_int64))] * 4
        L.tm_smoother_apply.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.tm_smoother_rhs.argtypes = [C.c_void_p, _dp]
        L.tm_smoother_assemble_csr.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.tm_smoother_apply_reference_order.argtypes = [C.c_void_p, _dp, _dp]
        L.tm_smoother_row_kinds.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.tm_smoother_control_function.argtypes = [C.c_void_p, _dp]
        L.tm_smoother_profile.argtypes = [C.c_void_p, C.c_int]
        L.tm_smoother_profile_read.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.tm_smoother_queue_ordering.argtypes = [C.c_void_p]
        L.tm_smoother_inner.argtypes = [C.c_void_p]
        L.tm_csr_ilu0_probe.argtypes = [C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp, _dp]
        L.tm_plan_build.argtypes = [C.POINTER(tm_mesh_desc), C.POINTER(tm_plan_rows)]
        L.tm_plan_free.argtypes = [C.POINTER(tm_plan_rows)]
        L.tm_plan_free.restype = None
        L.tm_export_soa.argtypes = [_dp, C.c_uint64, C.c_uint64, _dp, _dp]
        L.tm_smoother_export_soa.argtypes = [C.c_void_p, C.c_uint64, _dp, _dp, _dp, _dp]
        L.tm_rccl_unique_id.argtypes = [C.c_char_p, C.c_void_p]
        L.tm_rccl_comm_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.tm_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.tm_rccl_comm_destroy.restype = None
        L.tm_rccl_hooks.argtypes = [C.c_void_p, C.POINTER(tm_mesh_desc), C.POINTER(C.c_int32), C.POINTER(tm_comm_hooks)]
        L.tm_rccl_hooks_for.argtypes = [C.c_void_p, C.POINTER(tm_mesh_desc), C.POINTER(C.c_int32), C.POINTER(tm_solver_opt), C.POINTER(tm_control_fn), C.POINTER(tm_comm_hooks)]
        L.tm_plan_local.argtypes = [C.POINTER(tm_mesh_desc), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(tm_plan_local_info)]
        L.tm_plan_local_free.argtypes = [C.POINTER(tm_plan_local_info)]
        L.tm_plan_local_free.restype = None
        L.tm_dev_tfi_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64] + [C.c_void_p] * 8 + [C.c_void_p]
        L.tm_dev_relax_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.c_void_p]
        L.tm_rccl_peer_table_build.argtypes = [C.POINTER(tm_mesh_desc), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(tm_rccl_peer_table)]
        L.tm_rccl_peer_table_free.argtypes = [C.POINTER(tm_rccl_peer_table)]
        L.tm_rccl_peer_table_free.restype = None
        L.tm_white_math_probe.argtypes = [_dp, _dp, C.c_uint64, _dp, _dp]
        L.tm_stream_probe.argtypes = [C.c_uint64, C.c_int32, _dp, _dp]
        if hasattr(L, "tm_tune_apply"):   # measurement build (TM_HIP_LIB=.../libtm_hip_dbg.so, tools/)
            L.tm_debug_null_hooks.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(tm_comm_hooks)]
            L.tm_diag_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
            L.tm_tune_fuse.argtypes = [C.c_int]
            L.tm_tune_apply.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise TmError(rc, lib().tm_last_error().decode())
    return rc


def f64ptr(a):
    return a.ctypes.data_as(_dp)


class MeshDesc:
    """Marshals a discrete.Mesh into a tm_mesh_desc; keeps the ctypes arrays alive."""

    def __init__(self, mesh, with_coordinates=True):
        blocks = mesh.blocks
        self._blocks = (tm_block * max(1, len(blocks)))()
        for k, b in enumerate(blocks):
            data = b.points.data
            if with_coordinates and data is not None:
                if data.dtype != np.float64 or not data.flags["C_CONTIGUOUS"]:
                    raise TmError(TM_E_ARG, "block coordinates must be C-contiguous float64")
                self._blocks[k] = tm_block(f64ptr(data), b.points.size[0], b.points.size[1])
            else:
                self._blocks[k] = tm_block(None, b.points.size[0], b.points.size[1])
        conns = mesh.connections
        self._conns = (tm_connection * max(1, len(conns)))()
        for k, c in enumerate(conns):
            tc = tm_connection()
            for s in range(2):
                r = c.ranges[s]
                tc.r[s] = tm_range(r.block, int(r.side), 0, r.start, r.end)
            tc.has_periodicity = 0 if c.periodicity is None else 1
            if c.periodicity is not None:
                tc.periodicity[0], tc.periodicity[1] = float(c.periodicity[0]), float(c.periodicity[1])
            self._conns[k] = tc
        bcs = mesh.boundary_conditions
        self._bcs = (tm_condition * max(1, len(bcs)))()
        for k, bc in enumerate(bcs):
            r = bc.range
            self._bcs[k] = tm_condition(tm_range(r.block, int(r.side), 0, r.start, r.end), int(bc.kind), 0)
        self.desc = tm_mesh_desc(self._blocks, len(blocks), self._conns, len(conns), self._bcs, len(bcs))

    def ref(self):
        return C.byref(self.desc)
