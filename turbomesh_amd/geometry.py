"""Mirror of reference src/core/geometry.zig: Line (splines are next-row N1, not built yet)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Line:
    """geometry.zig:17-41"""

    start: np.ndarray
    end: np.ndarray

    def interpolate(self, clustering):
        u = np.asarray(clustering, dtype=np.float64)
        if u[0] != 0.0 or u[-1] != 1.0:
            raise ValueError("clustering must run from 0 to exactly 1")
        s = np.asarray(self.start, dtype=np.float64)
        dx = np.asarray(self.end, dtype=np.float64) - s
        return s[None, :] + u[:, None] * dx[None, :]
