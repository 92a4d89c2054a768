"""Mirror of reference src/core/clustering.zig (host-side edge preparation, O(n) per edge)."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class Uniform:
    """clustering.zig:9-17"""

    def compute(self, n):
        return np.arange(n, dtype=np.float64) / np.float64(n - 1)


@dataclass
class Roberts:
    """clustering.zig:24-42"""

    alpha: float
    beta: float

    def compute(self, n):
        out = np.empty(n)
        a, b = self.alpha, self.beta
        for i in range(n):
            u = float(i) / float(n - 1)
            tmp = math.pow((b + 1.0) / (b - 1.0), (u - a) / (1.0 - a))
            tbar = (b + 2.0 * a) * tmp - b + 2.0 * a
            out[i] = tbar / ((2.0 * a + 1.0) * (1.0 + tmp))
        return out


@dataclass
class SingleHyperbolicClustering:
    """clustering.zig:56-95 (Vinokur 1983, eq. 63-67)"""

    delta_s: float

    def compute(self, n):
        n_1 = float(n - 1)
        y = 1.0 / (n_1 * self.delta_s)
        if y < 2.7829681:
            yb = y - 1.0
            delta = math.sqrt(6.0 * yb) * (1.0 + yb * (-0.15 + yb * (0.057321429 + yb * (-0.024907295 + yb * (0.0077424461 - 0.0010794123 * yb)))))
        else:
            w = 1.0 / y - 0.028527431
            v = math.log(y)
            delta = v + (1.0 + 1.0 / v) * math.log(2.0 * v) - 0.02041793 + w * (0.24902722 + w * (1.9496443 + w * (-2.6294547 + 8.56795911 * w)))
        xi = np.arange(n, dtype=np.float64) / n_1
        out = xi.copy()
        for i in range(1, n):
            out[i] = 1.0 + math.tanh(0.5 * delta * (xi[i] - 1.0)) / math.tanh(0.5 * delta)
        return out


def create(clustering, n):
    """clustering.zig:110-116"""
    return clustering.compute(n)
