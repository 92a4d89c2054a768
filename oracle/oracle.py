"""ctypes front end of the CPU ORACLE (test infrastructure, NOT the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps oracle/libtm_oracle.so (built by `make -C oracle`), the C++ restatement
of turbomesh's TFI + elliptic smoothing path (see oracle/tm_oracle.h for the reference
file:line each function follows).

PARITY STATUS: "parity unpinned" -- the reference holds no golden vector for TFI or
smoothing; the adjacent KATs it does hold are checked in tests/test_oracle_kat.py.

A mesh is described duck-typed: an object with
    .blocks       list of float64 arrays shaped (ni, nj, 2), C-contiguous (mutated in place)
    .connections  list of (range0, range1, periodicity | None), range = (block, side, start, end)
    .bcs          list of (range, kind)
with side in 0..3 (i_min, i_max, j_min, j_max) and kind in 0..2 (wall, inlet, outlet).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtm_oracle.so")

SIDE_I_MIN, SIDE_I_MAX, SIDE_J_MIN, SIDE_J_MAX = 0, 1, 2, 3
BC_WALL, BC_INLET, BC_OUTLET = 0, 1, 2
CF_LAPLACE, CF_WHITE = 0, 1
SOLVER_GMRES, SOLVER_BICGSTAB, SOLVER_DIRECT, SOLVER_SCALED_BICGSTAB = 0, 1, 2, 4
PRECOND_DIAGONAL, PRECOND_ILU0 = 0, 1
KIND_FIXED, KIND_SMOOTHED, KIND_CONNECTED, KIND_LAPLACIAN, KIND_SLIDING = 0, 1, 2, 3, 4


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


class _Range(C.Structure):
    _fields_ = [("block", C.c_uint64), ("side", C.c_uint32), ("_pad", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64)]


class _Connection(C.Structure):
    _fields_ = [("r", _Range * 2), ("has_periodicity", C.c_int32), ("_pad", C.c_int32), ("periodicity", C.c_double * 2)]


class _Condition(C.Structure):
    _fields_ = [("range", _Range), ("kind", C.c_uint32), ("_pad", C.c_uint32)]


class _Block(C.Structure):
    _fields_ = [("xy", C.POINTER(C.c_double)), ("ni", C.c_uint64), ("nj", C.c_uint64)]


class _MeshDesc(C.Structure):
    _fields_ = [("blocks", C.POINTER(_Block)), ("nblocks", C.c_uint64), ("conns", C.POINTER(_Connection)), ("nconns", C.c_uint64),
                ("bcs", C.POINTER(_Condition)), ("nbcs", C.c_uint64)]


class _ControlFn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("ds_target", C.c_double), ("theta_target", C.c_double)]


class _SolverOpt(C.Structure):
    _fields_ = [("tag", C.c_int32), ("preconditioner", C.c_int32), ("rtol", C.c_double), ("atol", C.c_double), ("max_iters", C.c_uint64)]


class _Stats(C.Structure):
    _fields_ = [("outer_iterations", C.c_uint64), ("inner_iterations", C.c_uint64), ("last_residual", C.c_double),
                ("last_dx2", C.c_double), ("last_dy2", C.c_double), ("not_converged", C.c_int32), ("_pad", C.c_int32)]


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force: bool = False) -> str:
    """Compile the oracle (g++, -ffp-contract=off) if the .so is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp", ".h"))]
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_edge_combine_len.restype = C.c_uint64
        L.orc_system_create.restype = C.c_void_p
        L.orc_system_create.argtypes = [C.POINTER(_MeshDesc), C.POINTER(_ControlFn), C.POINTER(C.c_int)]
        for name in ("orc_system_destroy", "orc_system_seed_initial_guess"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        for name in ("orc_system_fill_x_specific", "orc_system_fill_y_specific"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_system_fill.argtypes = [C.c_void_p, C.c_uint64]
        for name in ("orc_system_dof", "orc_system_nnz", "orc_system_nboundary"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint64
        for name in ("orc_system_lhs_p", "orc_system_lhs_i", "orc_system_boundary_kind"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = _ip
        for name in ("orc_system_lhs_values", "orc_system_rhs_x", "orc_system_rhs_y", "orc_system_x_new", "orc_system_y_new",
                     "orc_system_control_function"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = _dp
        L.orc_system_commit.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_system_commit.restype = C.c_double
        L.orc_system_solve.argtypes = [C.c_void_p, C.POINTER(_SolverOpt), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        L.orc_system_matvec.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_system_matvec.restype = None
        L.orc_time_bicgstab_iterations.restype = C.c_double
        L.orc_time_bicgstab_iterations.argtypes = [C.c_uint64, C.c_uint64, _dp, C.c_uint64, _dp]
        L.orc_mirror_apply_block.argtypes = [C.c_int, C.c_uint64, C.c_uint64, _dp, _dp, _dp, _dp, C.c_double]
        L.orc_time_relax_sweeps.restype = C.c_double
        L.orc_time_relax_sweeps.argtypes = [C.c_uint64, C.c_uint64, _dp, _dp, C.c_uint64, C.c_double]
        L.orc_time_relax_sweeps_mt.restype = C.c_double
        L.orc_time_relax_sweeps_mt.argtypes = [C.c_uint64, C.c_uint64, _dp, _dp, C.c_uint64, C.c_double, C.c_uint32]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise OracleError(rc, lib().orc_last_error().decode())
    return rc


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


# ------------------------------------------------------------------ clustering / edges
def cluster_uniform(n):
    u = np.empty(n)
    lib().orc_cluster_uniform(u.ctypes.data_as(_dp), C.c_uint64(n))
    return u


def cluster_roberts(n, alpha, beta):
    u = np.empty(n)
    lib().orc_cluster_roberts(u.ctypes.data_as(_dp), C.c_uint64(n), C.c_double(alpha), C.c_double(beta))
    return u


def cluster_tanh(n, delta_s):
    u = np.empty(n)
    lib().orc_cluster_tanh(u.ctypes.data_as(_dp), C.c_uint64(n), C.c_double(delta_s))
    return u


def line_interpolate(start, end, u):
    u, up = _f64(u)
    out = np.empty((len(u), 2))
    s = (C.c_double * 2)(*start)
    e = (C.c_double * 2)(*end)
    lib().orc_line_interpolate(s, e, up, C.c_uint64(len(u)), out.ctypes.data_as(_dp))
    return out


def edge_combine(views):
    """views: list of (points[n,2], clustering[n], start, end) -- discrete.zig Edge.combine."""
    n = len(views)
    pts = [np.ascontiguousarray(v[0], dtype=np.float64) for v in views]
    cl = [np.ascontiguousarray(v[1], dtype=np.float64) for v in views]
    pp = (_dp * n)(*[p.ctypes.data_as(_dp) for p in pts])
    cp = (_dp * n)(*[c.ctypes.data_as(_dp) for c in cl])
    st = (C.c_uint64 * n)(*[v[2] for v in views])
    en = (C.c_uint64 * n)(*[v[3] for v in views])
    m = lib().orc_edge_combine_len(C.c_uint64(n), st, en)
    op = np.empty((m, 2))
    oc = np.empty(m)
    _check(lib().orc_edge_combine(C.c_uint64(n), pp, cp, st, en, op.ctypes.data_as(_dp), oc.ctypes.data_as(_dp)))
    return op, oc


# ------------------------------------------------------------------ TFI
def tfi_block(x_i_min, x_i_max, x_j_min, x_j_max, s1, s2, t1, t2):
    a, ap = _f64(x_i_min)
    b, bp = _f64(x_i_max)
    c, cp = _f64(x_j_min)
    d, dp = _f64(x_j_max)
    s1, s1p = _f64(s1)
    s2, s2p = _f64(s2)
    t1, t1p = _f64(t1)
    t2, t2p = _f64(t2)
    ni, nj = a.shape[0], c.shape[0]
    if b.shape[0] != ni or d.shape[0] != nj or len(s1) != ni or len(s2) != ni or len(t1) != nj or len(t2) != nj:
        raise OracleError(-1, "inconsistent edge sizes")
    out = np.empty((ni, nj, 2))
    _check(lib().orc_tfi_block(out.ctypes.data_as(_dp), C.c_uint64(ni), C.c_uint64(nj), ap, bp, cp, dp, s1p, s2p, t1p, t2p))
    return out


def tfi_linear2d(e_i_min, e_i_max, e_j_min, e_j_max):
    a, ap = _f64(e_i_min)
    b, bp = _f64(e_i_max)
    c, cp = _f64(e_j_min)
    d, dp = _f64(e_j_max)
    ni, nj = a.shape[0], c.shape[0]
    out = np.empty((ni, nj, 2))
    _check(lib().orc_tfi_linear2d(out.ctypes.data_as(_dp), C.c_uint64(ni), C.c_uint64(nj), ap, bp, cp, dp))
    return out


# ------------------------------------------------------------------ mesh marshalling
class _Marshalled:
    """Keeps the ctypes arrays alive for the duration of a call."""

    def __init__(self, mesh):
        blocks = mesh.blocks
        for b in blocks:
            if b.dtype != np.float64 or not b.flags["C_CONTIGUOUS"] or b.ndim != 3 or b.shape[2] != 2:
                raise OracleError(-6, "blocks must be C-contiguous float64 arrays shaped (ni, nj, 2)")
        self.blocks = (_Block * max(1, len(blocks)))()
        for k, b in enumerate(blocks):
            self.blocks[k] = _Block(b.ctypes.data_as(_dp), b.shape[0], b.shape[1])
        conns = list(mesh.connections)
        self.conns = (_Connection * max(1, len(conns)))()
        for k, (r0, r1, per) in enumerate(conns):
            c = _Connection()
            c.r[0] = _Range(r0[0], r0[1], 0, r0[2], r0[3])
            c.r[1] = _Range(r1[0], r1[1], 0, r1[2], r1[3])
            c.has_periodicity = 0 if per is None else 1
            if per is not None:
                c.periodicity[0], c.periodicity[1] = float(per[0]), float(per[1])
            self.conns[k] = c
        bcs = list(mesh.bcs)
        self.bcs = (_Condition * max(1, len(bcs)))()
        for k, (r, kind) in enumerate(bcs):
            self.bcs[k] = _Condition(_Range(r[0], r[1], 0, r[2], r[3]), kind, 0)
        self.desc = _MeshDesc(self.blocks, len(blocks), self.conns, len(conns), self.bcs, len(bcs))


def _cf(control):
    """control: None | 'laplace' | ('white', ds_target, theta_target)"""
    if control is None or control == "laplace":
        return _ControlFn(CF_LAPLACE, 0, 0.0, 0.0)
    return _ControlFn(CF_WHITE, 0, float(control[1]), float(control[2]))


@dataclass
class Stats:
    outer_iterations: int
    inner_iterations: int
    last_residual: float
    last_dx2: float
    last_dy2: float
    not_converged: int
    residual_history: np.ndarray


def smooth_mesh(mesh, iterations, solver=SOLVER_BICGSTAB, preconditioner=PRECOND_DIAGONAL, control=None, rtol=0.0, atol=0.0,
                max_iters=0):
    """smooth.zig:74-166 -- mutates mesh.blocks in place, returns Stats."""
    m = _Marshalled(mesh)
    opt = _SolverOpt(solver, preconditioner, rtol, atol, max_iters)
    cf = _cf(control)
    st = _Stats()
    hist = np.zeros(max(1, iterations))
    _check(lib().orc_smooth_mesh(C.byref(m.desc), C.c_uint64(iterations), C.byref(opt), C.byref(cf), C.byref(st),
                                 hist.ctypes.data_as(_dp)))
    return Stats(st.outer_iterations, st.inner_iterations, st.last_residual, st.last_dx2, st.last_dy2, st.not_converged,
                 hist[:iterations])


class System:
    """Stepping interface over RowCompressedMatrixSystem2d (smooth.zig:277-1166)."""

    def __init__(self, mesh, control=None):
        self._m = _Marshalled(mesh)
        self._mesh = mesh
        err = C.c_int(0)
        cf = _cf(control)
        self._h = lib().orc_system_create(C.byref(self._m.desc), C.byref(cf), C.byref(err))
        if not self._h:
            _check(err.value if err.value < 0 else -6)
        self.dof = int(lib().orc_system_dof(self._h))
        self.nnz = int(lib().orc_system_nnz(self._h))

    def close(self):
        if self._h:
            lib().orc_system_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, fn, n, dtype=np.float64):
        ptr = fn(self._h)
        return np.ctypeslib.as_array(ptr, shape=(n,))

    def fill(self, iteration):
        _check(lib().orc_system_fill(self._h, C.c_uint64(iteration)))

    def fill_x_specific(self):
        _check(lib().orc_system_fill_x_specific(self._h))

    def fill_y_specific(self):
        _check(lib().orc_system_fill_y_specific(self._h))

    @property
    def lhs_p(self):
        return self._view(lib().orc_system_lhs_p, self.dof + 1)

    @property
    def lhs_i(self):
        return self._view(lib().orc_system_lhs_i, self.nnz)

    @property
    def lhs_values(self):
        return self._view(lib().orc_system_lhs_values, self.nnz)

    @property
    def rhs_x(self):
        return self._view(lib().orc_system_rhs_x, self.dof)

    @property
    def rhs_y(self):
        return self._view(lib().orc_system_rhs_y, self.dof)

    @property
    def x_new(self):
        return self._view(lib().orc_system_x_new, self.dof)

    @property
    def y_new(self):
        return self._view(lib().orc_system_y_new, self.dof)

    @property
    def control_function(self):
        return self._view(lib().orc_system_control_function, 2 * self.dof).reshape(self.dof, 2)

    @property
    def boundary_kind(self):
        return self._view(lib().orc_system_boundary_kind, int(lib().orc_system_nboundary(self._h)))

    def csr(self):
        """scipy CSR view of the currently filled matrix (copy)."""
        import scipy.sparse as sp

        return sp.csr_matrix((self.lhs_values.copy(), self.lhs_i.copy(), self.lhs_p.copy()), shape=(self.dof, self.dof))

    def seed_initial_guess(self):
        lib().orc_system_seed_initial_guess(self._h)

    def matvec(self, x):
        x, xp = _f64(x)
        out = np.empty(self.dof)
        lib().orc_system_matvec(self._h, xp, out.ctypes.data_as(_dp))
        return out

    def solve(self, solver=SOLVER_BICGSTAB, preconditioner=PRECOND_DIAGONAL, rtol=0.0, atol=0.0, max_iters=0):
        opt = _SolverOpt(solver, preconditioner, rtol, atol, max_iters)
        it = C.c_uint64(0)
        nc = C.c_int32(0)
        _check(lib().orc_system_solve(self._h, C.byref(opt), C.byref(it), C.byref(nc)))
        return int(it.value), int(nc.value)

    def commit(self):
        """residual + copy-back (smooth.zig:112-153); returns ((sx+sy)^2, sx, sy)."""
        dx2 = C.c_double(0)
        dy2 = C.c_double(0)
        r = lib().orc_system_commit(self._h, C.byref(dx2), C.byref(dy2))
        return float(r), dx2.value, dy2.value


def picard_exact(mesh, iterations, control=None, keep_iterates=False, permc_spec="COLAMD"):
    """Exact Picard iteration = the reference with its UMFPACK backend (umfpack.zig:18-24):
    oracle-assembled CSR, each component solved by an independent sparse LU (scipy splu).
    Mutates mesh.blocks in place.  Returns (residual_history, iterates|None).
    permc_spec: SuperLU's column ordering -- a different elimination order is a different exact solver in floating point; the
    distance between two of them is the floor any comparison with "the exact iterate" can be held to
    (tests/test_oracle_self_distance.py).  When fillYSpecific leaves the values as fillXSpecific did (no sliding rows) the x
    factorisation serves the y system as well -- the same arithmetic as factorising it again."""
    import scipy.sparse.linalg as spla

    s = System(mesh, control)
    hist, iterates = [], []
    for n in range(iterations):
        s.fill(n)
        s.fill_x_specific()
        vx = s.lhs_values.copy()
        lu = spla.splu(s.csr().tocsc(), permc_spec=permc_spec)
        x = lu.solve(s.rhs_x.copy())
        s.fill_y_specific()
        if not np.array_equal(vx, s.lhs_values):
            lu = spla.splu(s.csr().tocsc(), permc_spec=permc_spec)
        y = lu.solve(s.rhs_y.copy())
        del lu
        s.x_new[:] = x
        s.y_new[:] = y
        hist.append(s.commit()[0])
        if keep_iterates:
            iterates.append([b.copy() for b in mesh.blocks])
    s.close()
    return np.array(hist), (iterates if keep_iterates else None)


def picard_direct(mesh, iterations, control=None, keep_iterates=False):
    """The same Picard iteration with the oracle's own banded LU (orc_solvers.cpp, partial pivoting inside the band) as the
    exact solver -- a third elimination order, for meshes whose bandwidth it can hold (single blocks, strips)."""
    s = System(mesh, control)
    hist, iterates = [], []
    for n in range(iterations):
        s.fill(n)
        s.solve(SOLVER_DIRECT)
        hist.append(s.commit()[0])
        if keep_iterates:
            iterates.append([b.copy() for b in mesh.blocks])
    s.close()
    return np.array(hist), (iterates if keep_iterates else None)


# ------------------------------------------------------------------ stand-alone CSR solvers
def csr_ilu0(n, Ap, Ai, Ax, rhs=None):
    """ILU(0) of a CSR matrix in its own pattern (BiCGStab.zig:178-277) and, with rhs, M^-1 rhs (:384-422).  Returns (lu [nnz], out [n] or None)."""
    Ap = np.ascontiguousarray(Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Ax, Axp = _f64(Ax)
    lu = np.empty(len(Ax))
    out = None
    rp = None
    if rhs is not None:
        rhs, rp = _f64(rhs)
        out = np.empty(n)
    L = lib()
    L.orc_csr_ilu0.argtypes = [C.c_uint64, _ip, _ip, _dp, _dp, _dp, _dp]
    _check(L.orc_csr_ilu0(C.c_uint64(n), Ap.ctypes.data_as(_ip), Ai.ctypes.data_as(_ip), Axp, rp, lu.ctypes.data_as(_dp), None if out is None else out.ctypes.data_as(_dp)))
    return lu, out


def csr_solve(kind, n, Ap, Ai, Ax, b, x0=None, precond=PRECOND_DIAGONAL, restart=30, max_iters=1000, rtol=1e-6, atol=1e-8):
    Ap = np.ascontiguousarray(Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Ax, Axp = _f64(Ax)
    b, bp = _f64(b)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    it = C.c_uint64(0)
    L = lib()
    app, aip = Ap.ctypes.data_as(_ip), Ai.ctypes.data_as(_ip)
    if kind == "bicgstab":
        rc = L.orc_csr_bicgstab(C.c_uint64(n), app, aip, Axp, bp, x.ctypes.data_as(_dp), precond, C.c_uint64(max_iters),
                                C.c_double(rtol), C.c_double(atol), C.byref(it))
    elif kind == "gmres":
        rc = L.orc_csr_gmres(C.c_uint64(n), app, aip, Axp, bp, x.ctypes.data_as(_dp), precond, C.c_uint64(restart),
                             C.c_uint64(max_iters), C.c_double(rtol), C.c_double(atol), C.byref(it))
    elif kind == "direct":
        rc = L.orc_csr_direct(C.c_uint64(n), app, aip, Axp, bp, x.ctypes.data_as(_dp))
    else:
        raise ValueError(kind)
    _check(rc)
    return x, int(it.value), rc == 0


# ------------------------------------------------------------------ cpu_baseline timing
def time_bicgstab_iterations(xy, iters):
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    fill = C.c_double(0)
    t = lib().orc_time_bicgstab_iterations(C.c_uint64(xy.shape[0]), C.c_uint64(xy.shape[1]), xy.ctypes.data_as(_dp),
                                           C.c_uint64(iters), C.byref(fill))
    return float(t), fill.value


def time_reference_path(xy, bicg_iters, gmres_iters):
    """Per-stage seconds of ONE outer iteration of the reference's CPU path (smooth.zig:104-154) on one (ni, nj, 2) block,
    single thread: init, fill, BiCGStab-diagonal (x-system), ILU(0) factorisation, GMRES(30)+ILU(0) (y-system), residual +
    copy-back.  Works on a copy of xy."""
    w = np.array(xy, dtype=np.float64, order="C", copy=True)
    out = np.zeros(8)
    L = lib()
    L.orc_time_reference_path.restype = C.c_int
    rc = L.orc_time_reference_path(C.c_uint64(w.shape[0]), C.c_uint64(w.shape[1]), w.ctypes.data_as(_dp), C.c_uint64(bicg_iters),
                                   C.c_uint64(gmres_iters), out.ctypes.data_as(_dp))
    if rc != 0:
        raise RuntimeError("orc_time_reference_path failed")
    return {"init_s": out[0], "fill_s": out[1], "bicgstab_diag_s": out[2], "bicgstab_iterations": int(out[3]), "ilu0_factor_s": out[4],
            "gmres30_ilu0_s": out[5], "gmres_iterations": int(out[6]), "residual_copyback_s": out[7]}


MIRROR_RAW, MIRROR_SCALED, MIRROR_RESID, MIRROR_RELAX = 0, 1, 2, 3


def mirror_apply_block(mode, vec, xk, pq=None, omega=1.0, out=None):
    """Interior rows of one (ni, nj, 2) block in the DEVICE operation order (orc_mirror.cpp); perimeter of out untouched."""
    vec, vp = _f64(vec)
    xk, xp = _f64(xk)
    pqa = None
    pqp = None
    if pq is not None:
        pqa, pqp = _f64(pq)
    if out is None:
        out = np.zeros_like(vec)
    _check(lib().orc_mirror_apply_block(mode, C.c_uint64(vec.shape[0]), C.c_uint64(vec.shape[1]), vp, xp, pqp, out.ctypes.data_as(_dp), C.c_double(omega)))
    return out


def time_relax_sweeps_mt(xy, sweeps, threads, omega=1.0):
    """time_relax_sweeps with the rows of every sweep split over `threads` host threads (same bits)."""
    assert xy.dtype == np.float64 and xy.flags["C_CONTIGUOUS"]
    scratch = np.empty_like(xy)
    return float(lib().orc_time_relax_sweeps_mt(C.c_uint64(xy.shape[0]), C.c_uint64(xy.shape[1]), xy.ctypes.data_as(_dp),
                                                scratch.ctypes.data_as(_dp), C.c_uint64(sweeps), C.c_double(omega), C.c_uint32(threads)))


def time_relax_sweeps(xy, sweeps, omega=1.0):
    """Runs `sweeps` Jacobi elliptic sweeps in place on xy (ni,nj,2); returns seconds."""
    assert xy.dtype == np.float64 and xy.flags["C_CONTIGUOUS"]
    scratch = np.empty_like(xy)
    return float(lib().orc_time_relax_sweeps(C.c_uint64(xy.shape[0]), C.c_uint64(xy.shape[1]), xy.ctypes.data_as(_dp),
                                             scratch.ctypes.data_as(_dp), C.c_uint64(sweeps), C.c_double(omega)))


def ref_white_math(x, y):
    """acos(x), atan2(y, x) with the reference's libm algorithm (orc_refmath.hpp) -> (acos, atan2) arrays."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    a, t = np.empty_like(x), np.empty_like(x)
    lib().orc_ref_white_math(x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), C.c_uint64(x.size), a.ctypes.data_as(_dp), t.ctypes.data_as(_dp))
    return a, t
