"""Mirror of reference src/core/smoothing/wall_control_function.zig: Algorithm = laplace | white."""
from __future__ import annotations

import math
from dataclasses import dataclass

from .. import _capi


@dataclass
class White:
    """wall_control_function.zig:56-68"""

    ds_target: float
    theta_target: float = 0.5 * math.pi


@dataclass
class Algorithm:
    """wall_control_function.zig:16-20: union(enum){laplace: void, white: White}"""

    white: "White | None" = None

    @classmethod
    def laplace(cls):
        return cls(None)

    def c_struct(self):
        if self.white is None:
            return _capi.tm_control_fn(_capi.TM_CF_LAPLACE, 0, 0.0, 0.0)
        return _capi.tm_control_fn(_capi.TM_CF_WHITE, 0, self.white.ds_target, self.white.theta_target)
