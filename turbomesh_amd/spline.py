"""Mirror of reference src/core/spline.zig: FittingSpline -- natural cubic spline parameterised by
(approximate) arc length over [0, 1] (chord-length parameters + a 200-sample arc-length table).
Host-side preprocessing of <= 500 blade points (SURVEY.md "next" row N1)."""
from __future__ import annotations

import math

import numpy as np


class FittingSpline:
    sample_count = 200   # spline.zig:22

    def __init__(self, points, degree=3):
        """spline.zig:24-64"""
        if degree != 3:
            raise ValueError("error.UnsupportedDegree")
        pts = np.array(points, dtype=np.float64)
        if len(pts) < 2:
            raise ValueError("error.NotEnoughPoints")
        self.points = pts
        self.dim = pts.shape[1]
        self.params, total_chord = self._chord_params(pts)
        self.second_derivs = [self._second_derivs(self.params, pts[:, d]) for d in range(self.dim)]
        self.total_length = total_chord
        self._build_arc_length_table()

    # spline.zig:145-160
    @staticmethod
    def _chord_params(pts):
        n = len(pts)
        params = np.zeros(n)
        total = 0.0
        for i in range(1, n):
            total += FittingSpline._distance(pts[i - 1], pts[i])
            params[i] = total
        if total == 0.0:
            return np.arange(n, dtype=np.float64) / float(n - 1), 0.0
        return params / total, total

    # spline.zig:162-206 (natural spline, Thomas algorithm)
    @staticmethod
    def _second_derivs(params, y):
        n = len(params)
        z = np.zeros(n)
        if n == 2:
            return z
        tmp = np.zeros(n)
        for i in range(1, n - 1):
            h_im1 = params[i] - params[i - 1]
            h_i = params[i + 1] - params[i]
            if h_im1 == 0.0 or h_i == 0.0:
                raise ValueError("error.CoincidentParameters")
            dy_im1 = y[i] - y[i - 1]
            dy_i = y[i + 1] - y[i]
            alpha = (dy_i / h_i) - (dy_im1 / h_im1)
            denom = 2.0 * (params[i + 1] - params[i - 1]) - h_im1 * tmp[i - 1]
            tmp[i] = h_i / denom
            z[i] = (6.0 * alpha - h_im1 * z[i - 1]) / denom
        z[n - 1] = 0.0
        for k in range(n - 2, -1, -1):
            z[k] = z[k] - tmp[k] * z[k + 1]
        return z

    @staticmethod
    def _distance(a, b):   # spline.zig:224-231
        s = 0.0
        for i in range(len(a)):
            d = b[i] - a[i]
            s += d * d
        return math.sqrt(s)

    def eval(self, param):
        """spline.zig:208-222"""
        u = min(max(param, 0.0), 1.0)
        idx = 0
        p = self.params
        while idx + 1 < len(p) and p[idx + 1] < u:
            idx += 1
        if idx >= len(p) - 1:
            idx = len(p) - 2
        h = p[idx + 1] - p[idx]
        a = (p[idx + 1] - u) / h
        b = (u - p[idx]) / h
        out = np.empty(self.dim)
        for d in range(self.dim):
            y0, y1 = self.points[idx, d], self.points[idx + 1, d]
            z0, z1 = self.second_derivs[d][idx], self.second_derivs[d][idx + 1]
            out[d] = a * y0 + b * y1 + ((a * a * a - a) * z0 + (b * b * b - b) * z1) * (h * h) / 6.0
        return out

    def _build_arc_length_table(self):
        """spline.zig:87-111"""
        n = self.sample_count + 1
        self.sample_params = np.arange(n, dtype=np.float64) / float(n - 1)
        self.sample_arc = np.zeros(n)
        length = 0.0
        prev = self.eval(self.sample_params[0])
        for i in range(1, n):
            cur = self.eval(self.sample_params[i])
            length += self._distance(prev, cur)
            self.sample_arc[i] = length
            prev = cur
        self.total_length = length
        if length == 0.0:
            self.sample_arc[:] = 0.0
            return
        self.sample_arc = self.sample_arc / length

    def _param_at_arc_fraction(self, u):
        """spline.zig:113-143"""
        if self.total_length == 0.0:
            return 0.0
        target = min(max(u, 0.0), 1.0)
        lo, hi = 0, len(self.sample_arc) - 1
        while lo < hi:
            mid = (lo + hi) // 2
            if self.sample_arc[mid] < target:
                lo = mid + 1
            else:
                hi = mid
        if lo == 0:
            return self.sample_params[0]
        a0, a1 = self.sample_arc[lo - 1], self.sample_arc[lo]
        p0, p1 = self.sample_params[lo - 1], self.sample_params[lo]
        t = (target - a0) / (a1 - a0) if a1 > a0 else 0.0
        return p0 + t * (p1 - p0)

    def interpolate(self, u):
        """spline.zig:74-81: points at the arc-length fractions u."""
        return np.array([self.eval(self._param_at_arc_fraction(float(v))) for v in u], dtype=np.float64)

    def integrate(self):
        return self.total_length
