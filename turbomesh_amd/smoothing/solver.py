"""Mirror of reference src/core/smoothing/solver.zig: Tag, Option -- plus the new `hip` member.

The reference's own backends (gmres, bicgstab, umfpack, petsc) are CPU code of the Zig
program; this package serves only `hip`.  Asking it for another tag raises
error.ExternalSolverNotEnabled, the reference's answer for a backend that was not built in
(solver.zig:48, 56)."""
from __future__ import annotations

import enum
from dataclasses import dataclass

from .. import _capi


class Preconditioner(enum.IntEnum):
    """preconditioner.zig:1-4"""

    diagonal = 0
    ilu0 = 1


class Tag(enum.IntEnum):
    """solver.zig:10-15 + hip"""

    gmres = 0
    bicgstab = 1
    umfpack = 2
    petsc = 3
    hip = 4


class Inner(enum.IntEnum):
    bicgstab = _capi.TM_INNER_BICGSTAB   # Picard + matrix-free BiCGStab on D^-1 A
    relax = _capi.TM_INNER_RELAX         # one fused Jacobi elliptic sweep per outer iteration
    mg_bicgstab = _capi.TM_INNER_MG_BICGSTAB   # bicgstab, right-preconditioned by one multigrid V-cycle per block
    gmres = _capi.TM_INNER_GMRES         # Picard + restarted GMRES(30), diagonal left preconditioner (GMRES.zig:300-423 on the device)
    auto = _capi.TM_INNER_AUTO           # mg_bicgstab when the largest block has >= 100 000 nodes (>= 1000 when no connection couples the blocks) and no block's cell aspect ratio varies strongly (boundary-layer clustering), bicgstab otherwise (decided at create)


@dataclass
class Option:
    """solver.zig:18-27 as a tagged record; `hip` carries the device solver's knobs."""

    tag: Tag = Tag.hip
    preconditioner: Preconditioner = Preconditioner.diagonal   # payload of gmres / bicgstab
    inner: Inner = Inner.bicgstab
    rtol: float = 0.0          # 0 -> library default: 7.5e-9 / nodes within [1e-16, 1e-14] (scaled residual)
    atol: float = 0.0
    max_inner: int = 0         # 0 -> max(10000, 12 sqrt(nodes)) (the reference: 1000, BiCGStab.zig:19, with its looser stop test)
    check_every: int = 0       # 0 -> 8
    omega: float = 0.0         # 0 -> 1.0
    single_sweep: bool = False # relax: one kernel pass per sweep (default: three per pass with fixed walls, two on coupled blocks)
    rtol_initial: bool = False   # Krylov modes: rtol relative to the initial residual of each inner solve (inexact Picard; rtol 0 -> 1e-2)
    eager_scalars: bool = False  # Krylov modes: the textbook launch sequence (a kernel per vector update, a launch per scalar step); default: two fused kernels per iteration

    @classmethod
    def hip(cls, **kw):
        return cls(tag=Tag.hip, **kw)

    def served_by_hip(self):
        """The hip option that honours this one -- what `--hip file` of the front end does with the solver an input file names:
        gmres -> Inner.gmres (GMRES(30) on the device), bicgstab -> Inner.bicgstab, the direct backends (umfpack, petsc: exact solves,
        umfpack.zig:18-24) -> Inner.auto with the library's tight default tolerance.  The reference's ILU(0) preconditioner
        (BiCGStab.zig:178-277) is a sequential recurrence with no device counterpart: the diagonal takes its place (a preconditioner
        changes the route, not the Picard iterate).  Returns (option, note or None)."""
        if self.tag == Tag.hip:
            return self, None
        inner = {Tag.gmres: Inner.gmres, Tag.bicgstab: Inner.bicgstab}.get(self.tag, Inner.auto)
        note = None
        if self.tag in (Tag.gmres, Tag.bicgstab) and self.preconditioner == Preconditioner.ilu0:
            note = "preconditioner ilu0 has no device counterpart: diagonal scaling is used (same Picard iterates, more inner iterations)"
        elif self.tag not in (Tag.gmres, Tag.bicgstab):
            note = f"direct solver `{self.tag.name}` is served by the iterative device solve at its tight default tolerance"
        return Option.hip(inner=inner), note

    def c_struct(self):
        # preconditioner: the payload of the reference's gmres / bicgstab options; with the hip tag it selects ILU(0) in the linear-solver slot
        # (Solver / tm_csr_solve, TM_OPT_PRECOND_ILU0) -- the matrix-free smoother refuses it (no assembled matrix to factorise)
        ilu = 8 if (self.tag == Tag.hip and self.preconditioner == Preconditioner.ilu0) else 0
        return _capi.tm_solver_opt(int(self.tag), int(self.inner), self.rtol, self.atol, self.max_inner, self.check_every,
                                   (1 if self.single_sweep else 0) | (2 if self.eager_scalars else 0) | (4 if self.rtol_initial else 0) | ilu, self.omega)


class Solver:
    """solver.zig:29-93 for the `hip` tag: holds the assembled system BY VALUE like the reference's backends (views of the
    caller's arrays: lhs_p, lhs_i, lhs_values, rhs_x, rhs_y, x_new, y_new -- smooth.zig:277-307) and solves both components on
    the MI355X through tm_csr_solve (seam 2).  `system` is any object with those attributes plus fillXSpecific() / fillYSpecific()
    (smooth.zig:1115-1165); like umfpack.zig:18-24 they are called before the respective values are taken."""

    def __init__(self, option: Option, system):
        if option.tag != Tag.hip:
            raise _capi.TmError(_capi.TM_E_UNSUPPORTED, "ExternalSolverNotEnabled: only the `hip` solver is built in (solver.zig:48, 56)")
        self.option = option
        self.system = system
        self.stats = None

    @classmethod
    def init(cls, option: Option, system) -> "Solver":
        return cls(option, system)

    def deinit(self):
        self.system = None

    def solve(self):
        import ctypes as C

        import numpy as np

        s = self.system
        s.fillXSpecific()
        ax_x = np.array(s.lhs_values, dtype=np.float64, copy=True)
        s.fillYSpecific()
        ax_y = np.ascontiguousarray(s.lhs_values, dtype=np.float64)
        p = np.ascontiguousarray(s.lhs_p, dtype=np.int32)
        i = np.ascontiguousarray(s.lhs_i, dtype=np.int32)
        n = len(p) - 1
        ip = C.POINTER(C.c_int32)
        opt = self.option.c_struct()
        st = _capi.tm_stats()
        for v in (s.rhs_x, s.rhs_y, s.x_new, s.y_new):
            assert v.dtype == np.float64 and v.flags["C_CONTIGUOUS"] and len(v) == n
        rc = _capi.check(_capi.lib().tm_csr_solve(n, p.ctypes.data_as(ip), i.ctypes.data_as(ip), _capi.f64ptr(ax_x), _capi.f64ptr(ax_y), _capi.f64ptr(s.rhs_x),
                                                  _capi.f64ptr(s.rhs_y), _capi.f64ptr(s.x_new), _capi.f64ptr(s.y_new), C.byref(opt), C.byref(st)))
        self.stats = st.as_dict()
        return rc == 0
