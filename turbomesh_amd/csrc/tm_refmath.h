// acos / atan2 as the REFERENCE computes them.  The reference is Zig: std.math.acos / atan / atan2 are ports of musl's acos.c, atan.c,
// atan2.c (FreeBSD msun e_acos.c, s_atan.c, e_atan2.c -- the fdlibm algorithms, published constants), not of glibc's and not of
// ocml's; the three agree to an ulp, and wall_control_function.zig:282-473 (White) amplifies an ulp in (P,Q) a hundredfold over a
// few Picard iterations on T106.  So the device evaluates these two functions with the reference's algorithm: every operation
// below is an IEEE add / multiply / divide / sqrt in the order of the C source (the library is built with -ffp-contract=off), and
// oracle/orc_refmath.hpp is the same text for the CPU side -- the two agree bit for bit (tests/test_gpu_refmath.py).
// fdlibm notice: Copyright (C) 1993 by Sun Microsystems, Inc.  Developed at SunSoft, a Sun Microsystems, Inc. business.
// Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#if defined(__HIPCC__)
#define TM_RM_FN __host__ __device__ inline
#else
#define TM_RM_FN inline
#endif

namespace tm_refmath {

TM_RM_FN uint64_t bits(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof(u));
    return u;
}
TM_RM_FN double from_bits(uint64_t u) {
    double x;
    memcpy(&x, &u, sizeof(x));
    return x;
}

TM_RM_FN double acos_R(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}

TM_RM_FN double acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const uint64_t u = bits(x);
    const uint32_t hx = static_cast<uint32_t>(u >> 32), lx = static_cast<uint32_t>(u);
    const uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {   // |x| >= 1 or nan
        if (((ix - 0x3ff00000u) | lx) == 0) {
            if (hx >> 31) return 2 * pio2_hi;   // acos(-1) = pi (+ a tiny term that only raises inexact)
            return 0.0;
        }
        return 0.0 / (x - x);
    }
    if (ix < 0x3fe00000u) {   // |x| < 0.5
        if (ix <= 0x3c600000u) return pio2_hi;   // |x| < 2^-57
        return pio2_hi - (x - (pio2_lo - x * acos_R(x * x)));
    }
    if (hx >> 31) {   // x < -0.5
        const double z = (1.0 + x) * 0.5;
        const double s = sqrt(z);
        const double w = acos_R(z) * s - pio2_lo;
        return 2 * (pio2_hi - (s + w));
    }
    const double z = (1.0 - x) * 0.5;   // x > 0.5
    const double s = sqrt(z);
    const double df = from_bits(bits(s) & 0xffffffff00000000ull);
    const double c = (z - df * df) / (s + df);
    const double w = acos_R(z) * s + c;
    return 2 * (df + w);
}

TM_RM_FN double atan(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                           9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                           4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    uint32_t ix = static_cast<uint32_t>(bits(x) >> 32);
    const uint32_t sign = ix >> 31;
    ix &= 0x7fffffffu;
    int id;
    if (ix >= 0x44100000u) {   // |x| >= 2^66
        if (x != x) return x;
        const double z = atanhi[3] + 0x1p-120;
        return sign ? -z : z;
    }
    if (ix < 0x3fdc0000u) {   // |x| < 0.4375
        if (ix < 0x3e400000u) return x;   // |x| < 2^-27
        id = -1;
    } else {
        x = fabs(x);
        if (ix < 0x3ff30000u) {   // |x| < 1.1875
            if (ix < 0x3fe60000u) {   // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else {   // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else {
            if (ix < 0x40038000u) {   // |x| < 2.4375
                id = 2;
                x = (x - 1.5) / (1.0 + 1.5 * x);
            } else {   // 2.4375 <= |x| < 2^66
                id = 3;
                x = -1.0 / x;
            }
        }
    }
    const double z = x * x;
    const double w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const double r = atanhi[id] - (x * (s1 + s2) - atanlo[id] - x);
    return sign ? -r : r;
}

TM_RM_FN double atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    const uint64_t ux = bits(x), uy = bits(y);
    uint32_t ix = static_cast<uint32_t>(ux >> 32), iy = static_cast<uint32_t>(uy >> 32);
    const uint32_t lx = static_cast<uint32_t>(ux), ly = static_cast<uint32_t>(uy);
    if (((ix - 0x3ff00000u) | lx) == 0) return atan(y);   // x = 1.0
    const uint32_t m = ((iy >> 31) & 1u) | ((ix >> 30) & 2u);   // 2 * sign(x) + sign(y)
    ix &= 0x7fffffffu;
    iy &= 0x7fffffffu;
    if ((iy | ly) == 0) {   // y = 0
        switch (m) {
            case 0:
            case 1: return y;    // atan(+-0, +anything) = +-0
            case 2: return pi;   // atan(+0, -anything) = pi
            default: return -pi; // atan(-0, -anything) = -pi
        }
    }
    if ((ix | lx) == 0) return (m & 1u) ? -pi / 2 : pi / 2;   // x = 0
    if (ix == 0x7ff00000u) {   // x = INF
        if (iy == 0x7ff00000u) {
            switch (m) {
                case 0: return pi / 4;
                case 1: return -pi / 4;
                case 2: return 3 * pi / 4;
                default: return -3 * pi / 4;
            }
        } else {
            switch (m) {
                case 0: return 0.0;
                case 1: return -0.0;
                case 2: return pi;
                default: return -pi;
            }
        }
    }
    if (ix + (64u << 20) < iy || iy == 0x7ff00000u) return (m & 1u) ? -pi / 2 : pi / 2;   // |y/x| > 2^64
    double z;
    if ((m & 2u) && iy + (64u << 20) < ix) z = 0.0;   // |y/x| < 2^-64, x < 0
    else z = atan(fabs(y / x));
    switch (m) {
        case 0: return z;                   // atan(+,+)
        case 1: return -z;                  // atan(-,+)
        case 2: return pi - (z - pi_lo);    // atan(+,-)
        default: return (z - pi_lo) - pi;   // atan(-,-)
    }
}

}  // namespace tm_refmath
