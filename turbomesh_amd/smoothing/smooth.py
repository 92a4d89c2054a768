"""Mirror of reference src/core/smoothing/smooth.zig: mesh(mesh_data, iterations, solver_option,
control_function_algorithm) -- executed on the MI355X through libtm_hip.so.

`Smoother` is the persistent-handle form (coordinates stay in HBM between calls)."""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from .. import _capi
from . import solver as _solver
from . import wall_control_function as _wcf

log = logging.getLogger("smoothing")   # std.log.scoped(.smoothing), smooth.zig:58


def _log_sink(ctx, what, iteration, value):
    """The reference's two per-iteration lines (smooth.zig:105, 136-137) through Python's logging."""
    if what == 0:
        log.info("iteration: %d", iteration)
    else:
        log.info("\tresidual: %r", value)


_LOG_CB = _capi.LOG_FN(_log_sink)   # kept alive for the lifetime of the module


class per_iteration_log:
    """Context manager: route the per-iteration lines of every smoother call inside it to logging ("smoothing" logger).
    Costs one reduction + host round trip per outer iteration (see include/tm_hip.h tm_set_log)."""

    def __init__(self, enable=True):
        self.enable = enable

    def __enter__(self):
        if self.enable:
            _capi.lib().tm_set_log(_LOG_CB, None)
        return self

    def __exit__(self, *exc):
        if self.enable:
            _capi.lib().tm_set_log(_capi.LOG_FN(), None)


def mesh(mesh_data, iterations: int, solver_option: "_solver.Option | None" = None,
         control_function_algorithm: "_wcf.Algorithm | None" = None):
    """smooth.zig:74-166: mutates mesh_data.blocks[b].points.data in place; returns the stats.  Logs like the reference
    ("iteration: n", "\tresidual: r" per outer iteration, then the elapsed time) when the `smoothing` logger is at INFO."""
    opt = (solver_option or _solver.Option.hip()).c_struct()
    cf = (control_function_algorithm or _wcf.Algorithm.laplace()).c_struct()
    md = _capi.MeshDesc(mesh_data)
    st = _capi.tm_stats()
    with per_iteration_log(log.isEnabledFor(logging.INFO)):
        rc = _capi.check(_capi.lib().tm_smooth_mesh(md.ref(), iterations, C.byref(opt), C.byref(cf), C.byref(st)))
    if rc == _capi.TM_W_NOT_CONVERGED:
        log.warning("hip solve did not converge in %d of %d outer iterations", st.not_converged, st.outer_iterations)
    log.info("elapsed time for smoothing: %.2f s", st.seconds)
    return st.as_dict()


class Smoother:
    """tm_smoother_* handle: upload once, iterate on the device, download when needed."""

    def __init__(self, mesh_data, solver_option=None, control_function_algorithm=None, hooks=None, stream=None):
        self._mesh = mesh_data
        self._md = _capi.MeshDesc(mesh_data)
        opt = (solver_option or _solver.Option.hip()).c_struct()
        cf = (control_function_algorithm or _wcf.Algorithm.laplace()).c_struct()
        h = C.c_void_p()
        self._hooks = hooks
        _capi.check(_capi.lib().tm_smoother_create(self._md.ref(), C.byref(opt), C.byref(cf), C.byref(hooks) if hooks is not None else None,
                                                   C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.dof = int(_capi.lib().tm_smoother_dof(self._h))
        self.inner = _solver.Inner(int(_capi.lib().tm_smoother_inner(self._h)))   # what Inner.auto resolved to

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().tm_smoother_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def iterate(self, iterations: int):
        st = _capi.tm_stats()
        _capi.check(_capi.lib().tm_smoother_iterate(self._h, iterations, C.byref(st)))
        return st.as_dict()

    def write(self, filename, with_control_function=True):
        """smooth.zig:396-414 system.write: the coordinates resident on the device (+ P, Q planes)."""
        from .. import output

        output.write_smoother(self, filename, with_control_function)

    def iterate_until(self, scaled_residual_tol: float, max_iterations: int = 1000):
        """Iterate until the scaled nonlinear residual is <= tol.  Returns (reached, stats)."""
        st = _capi.tm_stats()
        rc = _capi.lib().tm_smoother_iterate_until(self._h, max_iterations, C.c_double(scaled_residual_tol), C.byref(st))
        if rc < 0:
            _capi.check(rc)
        return rc == 0, st.as_dict()

    def iterate_until_update(self, update_rms_tol: float, max_iterations: int = 1000):
        """Iterate until the last outer iteration moved the nodes by <= tol RMS (the reference's per-iteration quantity,
        smooth.zig:112-137).  Returns (reached, stats)."""
        st = _capi.tm_stats()
        rc = _capi.lib().tm_smoother_iterate_until_update(self._h, max_iterations, C.c_double(update_rms_tol), C.byref(st))
        if rc < 0:
            _capi.check(rc)
        return rc == 0, st.as_dict()

    def download(self):
        _capi.check(_capi.lib().tm_smoother_download(self._h, self._md.ref()))

    def upload(self):
        _capi.check(_capi.lib().tm_smoother_upload(self._h, self._md.ref()))

    def apply(self, vec, scaled=False):
        """out = A(X) vec (optionally row-equilibrated) for a (dof, 2) host array."""
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        assert vec.shape == (self.dof, 2)
        out = np.empty_like(vec)
        _capi.check(_capi.lib().tm_smoother_apply(self._h, _capi.f64ptr(vec), _capi.f64ptr(out), 1 if scaled else 0))
        return out

    def assemble_csr(self):
        """The reference's assembled system for the current device coordinates (tm_smoother_assemble_csr): (Ap, Ai, Ax_x, Ax_y)."""
        n = C.c_uint64(0)
        _capi.check(_capi.lib().tm_smoother_assemble_csr(self._h, None, None, None, None, 0, C.byref(n)))
        nnz = int(n.value)
        Ap, Ai = np.empty(self.dof + 1, dtype=np.int32), np.empty(nnz, dtype=np.int32)
        Ax, Ay = np.empty(nnz), np.empty(nnz)
        ip = C.POINTER(C.c_int32)
        _capi.check(_capi.lib().tm_smoother_assemble_csr(self._h, Ap.ctypes.data_as(ip), Ai.ctypes.data_as(ip), _capi.f64ptr(Ax), _capi.f64ptr(Ay), nnz, C.byref(n)))
        return Ap, Ai, Ax, Ay

    def apply_reference_order(self, vec):
        """out = A(X) vec through the assembled system, rows summed in CSR order (the reference's mat-vec, bit for bit)."""
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        assert vec.shape == (self.dof, 2)
        out = np.empty_like(vec)
        _capi.check(_capi.lib().tm_smoother_apply_reference_order(self._h, _capi.f64ptr(vec), _capi.f64ptr(out)))
        return out

    def rhs(self):
        out = np.empty((self.dof, 2))
        _capi.check(_capi.lib().tm_smoother_rhs(self._h, _capi.f64ptr(out)))
        return out

    def row_kinds(self):
        out = np.empty(self.dof, dtype=np.int32)
        _capi.check(_capi.lib().tm_smoother_row_kinds(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def profile(self, every=1):
        """Bracket every `every`-th launch of the dominant kernel with a hipEvent pair (0 / False = off)."""
        _capi.check(_capi.lib().tm_smoother_profile(self._h, int(every)))

    def profile_read(self):
        """(milliseconds summed over the bracketed launches, how many were bracketed, launches in total) since the last read."""
        ms = C.c_double(0)
        timed = C.c_uint64(0)
        n = C.c_uint64(0)
        _capi.check(_capi.lib().tm_smoother_profile_read(self._h, C.byref(ms), C.byref(timed), C.byref(n)))
        return ms.value, int(timed.value), int(n.value)

    QUEUE_ORDERING = {-1: "none", 0: "counters", 1: "events (TM_PAIR_SYNC=events)", 2: "events (several multi-rank handles in this process)",
                      3: "events (self-test: both streams on one hardware queue)"}

    def queue_ordering(self):
        """How the handle orders the two queues of a pipelined pass (include/tm_hip_diag.h): code and its meaning."""
        code = int(_capi.lib().tm_smoother_queue_ordering(self._h))
        return code, self.QUEUE_ORDERING.get(code, "?")

    def control_function(self):
        out = np.empty((self.dof, 2))
        _capi.check(_capi.lib().tm_smoother_control_function(self._h, _capi.f64ptr(out)))
        return out


def plan_rows(mesh_data):
    """Host-only: the perimeter-row table (tm_plan_build) as numpy arrays.  Works without a GPU."""
    md = _capi.MeshDesc(mesh_data, with_coordinates=False)
    rows = _capi.tm_plan_rows()
    _capi.check(_capi.lib().tm_plan_build(md.ref(), C.byref(rows)))
    try:
        n = int(rows.nrows)
        as_np = np.ctypeslib.as_array
        out = {
            "row": as_np(rows.row, (n,)).copy(), "kind": as_np(rows.kind, (n,)).copy(), "ncols": as_np(rows.ncols, (n,)).copy(),
            "cols": as_np(rows.cols, (n * 9,)).reshape(n, 9).copy(), "coef_x": as_np(rows.coef_x, (n * 9,)).reshape(n, 9).copy(),
            "coef_y": as_np(rows.coef_y, (n * 9,)).reshape(n, 9).copy(), "rhs": as_np(rows.rhs, (n * 2,)).reshape(n, 2).copy(),
            "slot": as_np(rows.slot, (n * 9,)).reshape(n, 9).copy(),
        }
    finally:
        _capi.lib().tm_plan_free(C.byref(rows))
    return out
