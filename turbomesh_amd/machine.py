"""Mirror of reference src/core/machine.zig: Geometry{pitch, profile}, Profile{down_part, up_part}."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .spline import FittingSpline


class Profile:
    """machine.zig:19-45: two natural cubic arc-length splines (pressure / suction side)."""

    def __init__(self, down, up):
        down, up = np.asarray(down, dtype=np.float64), np.asarray(up, dtype=np.float64)
        if not np.array_equal(down[0], up[0]):
            raise ValueError("error.NonMatchingLeadingEdge")
        if not np.array_equal(down[-1], up[-1]):
            raise ValueError("error.NonMatchingTrailingEdge")
        assert len(down) > 1 and down[0, 0] < down[-1, 0]
        self.down_part = FittingSpline(down, 3)
        self.up_part = FittingSpline(up, 3)


@dataclass
class Geometry:
    """machine.zig:8-15"""

    pitch: float
    profile: Profile
