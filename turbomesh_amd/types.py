"""Mirror of reference src/core/types.zig: Float = f64, Index = usize, Vec2d, Mat2d.

A Mat2d is a C-contiguous float64 array shaped (size[0], size[1], 2): node (i, j) at flat
index j + size[1]*i (types.zig:94-96), x,y interleaved (16 B per node), NaN-initialised
(types.zig:84-88).  The same memory is handed to the C-ABI as `double*`."""
from __future__ import annotations

import numpy as np

Float = np.float64
Index = int


def Vec2d(v0, v1):
    """types.zig:16-21 -- a pair of f64."""
    return np.array([v0, v1], dtype=np.float64)


def eqlApprox(a, b, tol):
    """types.zig:43-45"""
    return bool(abs(a[0] - b[0]) <= tol and abs(a[1] - b[1]) <= tol)


class Mat2d:
    """types.zig:78-101"""

    def __init__(self, size):
        self.size = (int(size[0]), int(size[1]))
        self.data = np.full((self.size[0], self.size[1], 2), np.nan, dtype=np.float64)

    @classmethod
    def placeholder(cls, size):
        """Sizes only (a block owned by another rank of a multi-GPU job): data is None."""
        m = cls.__new__(cls)
        m.size = (int(size[0]), int(size[1]))
        m.data = None
        return m

    @classmethod
    def from_array(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.ndim == 3 and arr.shape[2] == 2
        m = cls.__new__(cls)
        m.size = (arr.shape[0], arr.shape[1])
        m.data = arr
        return m

    def index(self, i):
        return i[1] + self.size[1] * i[0]

    def getIndex(self, i):
        return self.data[i[0], i[1]]
