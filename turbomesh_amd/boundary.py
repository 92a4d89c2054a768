"""Mirror of reference src/core/boundary.zig: Side, Range, Connection, Condition."""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Optional, Tuple


class Side(enum.IntEnum):
    """boundary.zig:8-13.  NOTE the edge naming: the i_min edge is parametrised by i at j = 0."""

    i_min = 0
    i_max = 1
    j_min = 2
    j_max = 3


class ConditionTag(enum.IntEnum):
    """boundary.zig:178-182"""

    wall = 0
    inlet = 1
    outlet = 2


@dataclass
class Range:
    """boundary.zig:15-26"""

    block: int
    side: Side
    start: int
    end: int

    def len(self):
        return abs(self.end - self.start) + 1


@dataclass
class Connection:
    """boundary.zig:119-139: periodicity maps ranges[0] onto ranges[1] (None = plain interface)."""

    ranges: Tuple[Range, Range]
    periodicity: Optional[Tuple[float, float]] = None

    def len(self):
        assert self.ranges[0].len() == self.ranges[1].len()
        return self.ranges[0].len()

    def lenInternal(self):
        return self.len() - 2


@dataclass
class Condition:
    """boundary.zig:184-187"""

    range: Range
    kind: ConditionTag
