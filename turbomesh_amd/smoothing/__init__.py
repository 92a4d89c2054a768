"""Mirror of reference src/core/smoothing/smoothing.zig."""
from . import smooth, solver, wall_control_function  # noqa: F401
from .solver import Preconditioner  # noqa: F401
