"""Mirror of reference src/core/input.zig: the JSON schema of the examples (Input), create_profile.
Tagged unions are single-key objects exactly as std.json parses them (examples/T106/T106.json).  The `solver`
key additionally accepts {"hip": {...}}."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import clustering as cluster
from . import csv as _csv
from .machine import Geometry, Profile
from .smoothing import solver as _solver
from .smoothing import wall_control_function as _wcf
from .templates.O4H import O4H, NumCells


def create_profile(profile_input: dict, scale: float, base_dir: str = ".") -> Profile:
    """input.zig:43-90"""
    if "data" in profile_input:
        down = np.array(profile_input["data"]["down"], dtype=np.float64)
        up = np.array(profile_input["data"]["up"], dtype=np.float64)
    else:
        down = _read_side(os.path.join(base_dir, profile_input["csv"]["down_csv_path"]))
        up = _read_side(os.path.join(base_dir, profile_input["csv"]["up_csv_path"]))
    if scale != 1.0:
        down = down * scale
        up = up * scale
    return Profile(down, up)


def _read_side(path):
    """input.zig:100-108: reverse if the first point lies downstream of the last."""
    side = _csv.parseCsvIntoVec2d(path)
    if side[0, 0] > side[-1, 0]:
        side = side[::-1].copy()
    return side


def _clustering(obj):
    (tag, payload), = obj.items()
    if tag == "uniform":
        return cluster.Uniform()
    if tag == "roberts":
        return cluster.Roberts(payload["alpha"], payload["beta"])
    if tag == "single_hyperbolic_clustering":
        return cluster.SingleHyperbolicClustering(payload["delta_s"])
    raise ValueError(f"unknown clustering {tag}")


@dataclass
class Input:
    """input.zig:25-41"""

    template: O4H
    iterations: int
    solver: "_solver.Option"
    wall_control_function: "_wcf.Algorithm"
    geometry_scale: float
    pitch: float
    profile: dict
    output: Optional[str] = None
    gui: Optional[bool] = None

    @classmethod
    def parse(cls, text: str) -> "Input":
        j = json.loads(text)
        (tname, t), = j["template"].items()
        if tname != "O4H":
            raise ValueError(f"unknown template {tname}")
        tmpl = O4H(_clustering(t["blade_clustering"]), NumCells(**t["num_cells"]), t.get("inlet_distance"), t.get("outlet_distance"))
        sm = j["smoothing"]
        (stag, spay), = sm["solver"].items()
        if stag == "hip":
            opt = _solver.Option.hip(**{k: (getattr(_solver.Inner, v) if k == "inner" else v) for k, v in spay.items()})
        else:
            opt = _solver.Option(tag=getattr(_solver.Tag, stag), preconditioner=getattr(_solver.Preconditioner, spay.get("preconditioner", "diagonal")))
        wcf = sm.get("wall_control_function", {"laplace": {}})
        (wtag, wpay), = wcf.items()
        algo = _wcf.Algorithm.laplace() if wtag == "laplace" else _wcf.Algorithm(_wcf.White(wpay["ds_target"], wpay.get("theta_target", 1.5707963267948966)))
        g = j["geometry"]
        return cls(tmpl, sm.get("iterations", 0), opt, algo, g.get("scale", 1.0), g["pitch"], g["profile"], j.get("output"), j.get("gui"))

    def geometry(self, base_dir="."):
        """gui/main.zig:42-45: Geometry.init(scale * pitch, create_profile(profile, scale))"""
        return Geometry(self.geometry_scale * self.pitch, create_profile(self.profile, self.geometry_scale, base_dir))
