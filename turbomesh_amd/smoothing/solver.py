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


@dataclass
class Option:
    """solver.zig:18-27 as a tagged record; `hip` carries the device solver's knobs."""

    tag: Tag = Tag.hip
    preconditioner: Preconditioner = Preconditioner.diagonal   # payload of gmres / bicgstab
    inner: Inner = Inner.bicgstab
    rtol: float = 0.0          # 0 -> library default 1e-12 (scaled residual)
    atol: float = 0.0
    max_inner: int = 0         # 0 -> 1000 (BiCGStab.zig:19)
    check_every: int = 0       # 0 -> 8
    omega: float = 0.0         # 0 -> 1.0
    single_sweep: bool = False # relax: one kernel pass per sweep (default: two sweeps per pass where possible)

    @classmethod
    def hip(cls, **kw):
        return cls(tag=Tag.hip, **kw)

    def c_struct(self):
        return _capi.tm_solver_opt(int(self.tag), int(self.inner), self.rtol, self.atol, self.max_inner, self.check_every,
                                   1 if self.single_sweep else 0, self.omega)
