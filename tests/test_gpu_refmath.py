"""The two libm functions of the White control function (wall_control_function.zig:282-473): acos and atan2.

The reference is Zig, whose std.math.acos / atan / atan2 are ports of musl (fdlibm's e_acos.c, s_atan.c, e_atan2.c).  Device and
oracle both evaluate that algorithm (turbomesh_amd/csrc/tm_refmath.h, oracle/orc_refmath.hpp) instead of ocml's / glibc's, so
that (P, Q) carry no libm difference between the two sides -- and, as far as can be said without a Zig toolchain, none against the
reference either."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle


def _inputs():
    rng = np.random.default_rng(7)
    n = 400000
    x = np.concatenate([rng.uniform(-1, 1, n), 1.0 - 10.0 ** rng.uniform(-16, 0, n // 4), -1.0 + 10.0 ** rng.uniform(-16, 0, n // 4),
                        [1.0, -1.0, 0.0, -0.0, 0.5, -0.5, 1e-20, -1e-20, 2.0 ** -57, 0.4999999999999999, 0.5000000000000001]])
    m = x.size
    y = rng.normal(size=m) * 10.0 ** rng.integers(-12, 12, m)
    xs = rng.normal(size=m) * 10.0 ** rng.integers(-12, 12, m)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e300, -1e300, 1e-300, 5e-324, 3.0, -3.0])
    k = special.size
    y[:k * k] = np.repeat(special, k)
    xs[:k * k] = np.tile(special, k)
    return x, y, xs


def test_oracle_refmath_is_within_one_ulp_of_the_host_libm():
    # a wrong digit in any of the 33 constants or a wrong branch would show as more than an ulp
    x, y, xs = _inputs()
    a, _ = oracle.ref_white_math(x, y)
    _, t = oracle.ref_white_math(xs, y)
    with np.errstate(invalid="ignore"):
        for got, ref in ((a, np.arccos(x)), (t, np.arctan2(y, xs))):
            ok = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), ok)
            ulp = np.spacing(np.maximum(np.abs(ref[ok]), np.abs(got[ok])))
            assert np.all(np.abs(got[ok] - ref[ok]) <= ulp)
            assert np.array_equal(np.signbit(got[ok]), np.signbit(ref[ok]))
    # known values of the algorithm's own special cases (musl acos.c / atan2.c)
    a, t = oracle.ref_white_math(np.array([1.0, -1.0, 0.0]), np.array([0.0, 0.0, 1.0]))
    assert a[0] == 0.0 and a[1] == np.pi and a[2] == np.pi / 2
    assert t[0] == 0.0 and t[1] == np.pi and t[2] == np.pi / 2


@pytest.mark.gpu
def test_device_and_oracle_refmath_agree_bit_for_bit():
    from turbomesh_amd import _capi

    L = _capi.lib()
    dp = C.POINTER(C.c_double)
    L.tm_white_math_probe.argtypes = [dp, dp, C.c_uint64, dp, dp]
    L.tm_white_math_probe.restype = C.c_int
    x, y, xs = _inputs()
    for xin in (x, xs):
        xin = np.ascontiguousarray(xin)
        a = np.empty_like(xin)
        t = np.empty_like(xin)
        _capi.check(L.tm_white_math_probe(xin.ctypes.data_as(dp), y.ctypes.data_as(dp), xin.size, a.ctypes.data_as(dp), t.ctypes.data_as(dp)))
        ra, rt = oracle.ref_white_math(xin, y)
        assert np.array_equal(a.view(np.uint64)[~np.isnan(ra)], ra.view(np.uint64)[~np.isnan(ra)])
        assert np.array_equal(np.isnan(a), np.isnan(ra))
        assert np.array_equal(t.view(np.uint64)[~np.isnan(rt)], rt.view(np.uint64)[~np.isnan(rt)])
        assert np.array_equal(np.isnan(t), np.isnan(rt))
