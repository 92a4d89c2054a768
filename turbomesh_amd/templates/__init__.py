"""Mirror of reference src/core/templates/templates.zig."""
from .O4H import O4H, NumCells  # noqa: F401
