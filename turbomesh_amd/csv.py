"""Mirror of reference src/core/csv.zig: space-separated `x y` reader, lines starting with # are comments."""
from __future__ import annotations

import numpy as np


def parseCsvIntoVec2d(file_path):
    """csv.zig:10-57"""
    rows = []
    with open(file_path, "r") as f:
        for i_line, line in enumerate(f.read().split("\n")):
            if line == "":
                continue
            if line[0] == "#":
                continue
            entries = [e for e in line.split(" ") if e != ""]
            if len(entries) > 2:
                raise ValueError(f"csv parsing error: too many entries in line detected\n  file: {file_path}\n  line {i_line}: {line}")
            if len(entries) != 2:
                raise ValueError("csv read error")
            rows.append((float(entries[0]), float(entries[1])))
    return np.array(rows, dtype=np.float64)
