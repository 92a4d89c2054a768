"""turbomesh_amd -- MI355X-native structured-grid smoother behind turbomesh's block/mesh API.

Host-side mirror of the reference's `core` module for the hot path only (reference
src/core/lib.zig): types, clustering, geometry.Line, boundary, discrete, tfi, smoothing.
All compute goes through the C-ABI of libtm_hip.so (include/tm_hip.h); there is no CPU
fallback."""
from . import boundary, clustering, discrete, geometry, smoothing, tfi, types  # noqa: F401
from ._capi import TmError  # noqa: F401

__all__ = ["boundary", "clustering", "discrete", "geometry", "smoothing", "tfi", "types", "TmError"]
