"""Mirror of reference src/core/tfi.zig -- computed on the MI355X through libtm_hip.so.

linear2dBoundaryBlendedControlFunction keeps the reference's name, argument order and
meaning (tfi.zig:112-122); the debug asserts of the reference surface as TmError."""
from __future__ import annotations

import numpy as np

from . import _capi
from .types import Mat2d


def _edge(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise _capi.TmError(_capi.TM_E_SIZE, "an edge is an (n, 2) array of Vec2d")
    return a


def linear2dBoundaryBlendedControlFunction(data: Mat2d, x_i_min, x_i_max, x_j_min, x_j_max, s1, s2, t1, t2) -> None:
    """tfi.zig:112-208: boundary-blended linear TFI, overwrites every node of `data`."""
    a, b, c, d = _edge(x_i_min), _edge(x_i_max), _edge(x_j_min), _edge(x_j_max)
    s1, s2, t1, t2 = (np.ascontiguousarray(v, dtype=np.float64) for v in (s1, s2, t1, t2))
    n, m = a.shape[0], c.shape[0]
    if b.shape[0] != n or len(s1) != n or len(s2) != n or d.shape[0] != m or len(t1) != m or len(t2) != m or data.size != (n, m):
        raise _capi.TmError(_capi.TM_E_SIZE, "edge / clustering / block sizes do not agree (tfi.zig:125-133)")
    f = _capi.f64ptr
    _capi.check(_capi.lib().tm_tfi_block(f(data.data), n, m, f(a), f(b), f(c), f(d), f(s1), f(s2), f(t1), f(t2)))


def linear2d(data: np.ndarray, edge_i_min, edge_i_max, edge_j_min, edge_j_max) -> None:
    """tfi.zig:19-67: plain linear TFI into a (ni, nj, 2) array."""
    a, b, c, d = _edge(edge_i_min), _edge(edge_i_max), _edge(edge_j_min), _edge(edge_j_max)
    n, m = a.shape[0], c.shape[0]
    if b.shape[0] != n or d.shape[0] != m:
        raise _capi.TmError(_capi.TM_E_SIZE, "error.InconsistentSize (tfi.zig:30)")
    if data.shape != (n, m, 2) or data.dtype != np.float64 or not data.flags["C_CONTIGUOUS"]:
        raise _capi.TmError(_capi.TM_E_SIZE, "output must be a C-contiguous float64 (ni, nj, 2) array")
    f = _capi.f64ptr
    _capi.check(_capi.lib().tm_tfi_linear2d(f(data), n, m, f(a), f(b), f(c), f(d)))
