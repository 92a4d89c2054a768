// ORACLE (test infrastructure) -- clustering, Edge.combine, Line.interpolate and the two TFI
// variants of the reference.  Operation order is kept term by term so that a build with
// -ffp-contract=off reproduces the reference's IEEE-754 results (Zig does not contract a*b+c).
#include "orc_types.hpp"
#include "tm_oracle.h"
#include <cstring>

namespace orc {

// clustering.zig:9-17
void cluster_uniform(Float* data, Index n) {
    for (Index i = 0; i < n; ++i) data[i] = static_cast<Float>(i) / static_cast<Float>(n - 1);
}

// clustering.zig:24-42
void cluster_roberts(Float* data, Index n, Float alpha, Float beta) {
    for (Index i = 0; i < n; ++i) {
        const Float u = static_cast<Float>(i) / static_cast<Float>(n - 1);
        const Float tmp = std::pow((beta + 1.0) / (beta - 1.0), (u - alpha) / (1.0 - alpha));
        const Float tbar = (beta + 2.0 * alpha) * tmp - beta + 2.0 * alpha;
        data[i] = tbar / ((2.0 * alpha + 1.0) * (1.0 + tmp));
    }
}

// clustering.zig:56-95 (Vinokur 1983, eq. 63-67)
void cluster_tanh(Float* data, Index n, Float delta_s) {
    const Float n_1 = static_cast<Float>(n - 1);
    const Float b = n_1 * delta_s;
    const Float y = 1.0 / b;
    Float delta;
    if (y < 2.7829681) {
        const Float y_bar = y - 1.0;
        delta = std::sqrt(6.0 * y_bar) *
                (1.0 + y_bar * (-0.15 + y_bar * (0.057321429 + y_bar * (-0.024907295 + y_bar * (0.0077424461 - 0.0010794123 * y_bar)))));
    } else {
        const Float w = 1.0 / y - 0.028527431;
        const Float v = std::log(y);
        delta = v + (1.0 + 1.0 / v) * std::log(2.0 * v) - 0.02041793 +
                w * (0.24902722 + w * (1.9496443 + w * (-2.6294547 + 8.56795911 * w)));
    }
    for (Index i = 0; i < n; ++i) data[i] = static_cast<Float>(i) / n_1;
    for (Index i = 1; i < n; ++i) {
        const Float s = 1.0 + std::tanh(0.5 * delta * (data[i] - 1.0)) / std::tanh(0.5 * delta);
        data[i] = s;
    }
}

// geometry.zig:21-40
void line_interpolate(Vec2d start, Vec2d end, const Float* u, Index n, Vec2d* out) {
    const Vec2d dx = sub(end, start);
    for (Index i = 0; i < n; ++i) out[i] = add(start, scale(u[i], dx));
}

// discrete.zig:94-136 (EdgeView) + :38-91 (Edge.combine)
static Index view_len(Index start, Index end) { return start > end ? start - end + 1 : end - start + 1; }

Index edge_combine_len(Index nviews, const uint64_t* start, const uint64_t* end) {
    Index n = 0;
    for (Index v = 0; v < nviews; ++v) n += view_len(start[v], end[v]);
    n -= nviews - 1;
    return n;
}

int edge_combine(Index nviews, const Vec2d* const* points, const Float* const* clus, const uint64_t* start,
                 const uint64_t* end, Vec2d* out_points, Float* out_u) {
    const Float tol = 1e-10;
    for (Index v = 0; v + 1 < nviews; ++v) {   // discrete.zig:45-54
        if (!eqlApprox(points[v][end[v]], points[v + 1][start[v + 1]], tol)) return ORC_E_MISMATCH;
    }
    const Index n = edge_combine_len(nviews, start, end);
    // points: clonePoints, each view overwrites the previous view's last point (discrete.zig:67-70, 106-117)
    {
        Index pos = 0;
        for (Index v = 0; v < nviews; ++v) {
            const Index len = view_len(start[v], end[v]);
            if (start[v] > end[v]) {
                for (Index k = 0; k < len; ++k) out_points[pos + k] = points[v][start[v] - k];
            } else {
                for (Index k = 0; k < len; ++k) out_points[pos + k] = points[v][start[v] + k];
            }
            pos += len - 1;
        }
    }
    // clustering: cumulative sum of deltas relative to the view's first stored value, then
    // normalisation (discrete.zig:73-84, 119-135).  Note cloneClustering walks first..last in
    // STORAGE order even for reversed views and measures every delta from clustering[first].
    {
        Index pos = 0;
        Float last_value = 0.0;
        for (Index v = 0; v < nviews; ++v) {
            const Index first = start[v] > end[v] ? end[v] : start[v];
            const Index last = start[v] > end[v] ? start[v] : end[v];
            out_u[pos] = last_value;
            const Float base = clus[v][first];
            Index i_buf = 1;
            for (Index i = first + 1; i <= last; ++i) {
                const Float delta = clus[v][i] - base;
                out_u[pos + i_buf] = last_value + delta;
                i_buf += 1;
            }
            pos += i_buf - 1;
            last_value = out_u[pos];
        }
        for (Index i = 0; i < n; ++i) out_u[i] /= last_value;
    }
    return ORC_OK;
}

// types.zig:51-55 addAll over 4 terms: starts from (0,0) and adds left to right.
static inline Vec2d addAll4(Vec2d a, Vec2d b, Vec2d c, Vec2d d) {
    Vec2d res = vinit(0, 0);
    res = add(res, a);
    res = add(res, b);
    res = add(res, c);
    res = add(res, d);
    return res;
}
static inline Vec2d addAll3(Vec2d a, Vec2d b, Vec2d c) {
    Vec2d res = vinit(0, 0);
    res = add(res, a);
    res = add(res, b);
    res = add(res, c);
    return res;
}

// tfi.zig:112-208 linear2dBoundaryBlendedControlFunction
int tfi_block(Vec2d* data, Index n, Index m, const Vec2d* x_i_min, const Vec2d* x_i_max, const Vec2d* x_j_min,
              const Vec2d* x_j_max, const Float* s1, const Float* s2, const Float* t1, const Float* t2) {
    if (n < 2 || m < 2) return ORC_E_SIZE;
    // tfi.zig:135-145 (debug asserts in the reference)
    if (s1[0] != 0 || s1[n - 1] != 1.0 || s2[0] != 0 || s2[n - 1] != 1.0) return ORC_E_ARG;
    if (t1[0] != 0 || t1[m - 1] != 1.0 || t2[0] != 0 || t2[m - 1] != 1.0) return ORC_E_ARG;
    const Float tol = 1e-10;
    const Vec2d x_0_0 = x_i_min[0];
    const Vec2d x_n_0 = x_i_min[n - 1];
    const Vec2d x_0_m = x_j_min[m - 1];
    const Vec2d x_n_m = x_i_max[n - 1];
    if (!eqlApprox(x_0_0, x_j_min[0], tol) || !eqlApprox(x_n_0, x_j_max[0], tol) || !eqlApprox(x_0_m, x_i_max[0], tol) ||
        !eqlApprox(x_n_m, x_j_max[m - 1], tol))
        return ORC_E_MISMATCH;

    Index idx = 0;
    for (Index i = 0; i < n; ++i) {
        const Float s1_i = s1[i], s2_i = s2[i];
        const Vec2d x_i_0 = x_i_min[i], x_i_m = x_i_max[i];
        for (Index j = 0; j < m; ++j) {
            const Float t1_j = t1[j], t2_j = t2[j];
            const Vec2d x_0_j = x_j_min[j], x_n_j = x_j_max[j];
            const Float u = ((1.0 - t1_j) * s1_i + t1_j * s2_i) / (1.0 - (s2_i - s1_i) * (t2_j - t1_j));   // :185
            const Float v = ((1.0 - s1_i) * t1_j + s1_i * t2_j) / (1.0 - (t2_j - t1_j) * (s2_i - s1_i));   // :186
            const Vec2d u_ij = add(scale(1.0 - u, x_0_j), scale(u, x_n_j));
            const Vec2d v_ij = add(scale(1.0 - v, x_i_0), scale(v, x_i_m));
            const Vec2d uv_ij = addAll4(scale(u * v, x_n_m), scale(u * (1.0 - v), x_n_0), scale((1.0 - u) * v, x_0_m),
                                        scale((1.0 - u) * (1.0 - v), x_0_0));
            data[idx] = sub(add(u_ij, v_ij), uv_ij);
            idx += 1;
        }
    }
    return ORC_OK;
}

// tfi.zig:19-67 linear2d (corners from the i edges, uniform xi/eta)
int tfi_linear2d(Vec2d* data, Index n, Index m, const Vec2d* e_i_min, const Vec2d* e_i_max, const Vec2d* e_j_min,
                 const Vec2d* e_j_max) {
    if (n < 2 || m < 2) return ORC_E_SIZE;
    const Vec2d c00 = e_i_min[0], c10 = e_i_min[n - 1], c01 = e_i_max[0], c11 = e_i_max[n - 1];
    for (Index i = 0; i < n; ++i) {
        const Vec2d v_xi_0 = e_i_min[i], v_xi_1 = e_i_max[i];
        const Float xi = static_cast<Float>(i) / static_cast<Float>(n - 1);
        for (Index j = 0; j < m; ++j) {
            const Vec2d v_0_eta = e_j_min[j], v_1_eta = e_j_max[j];
            const Float eta = static_cast<Float>(j) / static_cast<Float>(m - 1);
            const Vec2d u_ij = add(scale(1.0 - xi, v_0_eta), scale(xi, v_1_eta));
            const Vec2d v_ij = add(scale(1.0 - eta, v_xi_0), scale(eta, v_xi_1));
            const Vec2d uv_ij = addAll4(scale(xi * eta, c11), scale(xi * (1.0 - eta), c10), scale((1.0 - xi) * eta, c01),
                                        scale((1.0 - xi) * (1.0 - eta), c00));
            data[i * m + j] = addAll3(u_ij, v_ij, negate(uv_ij));
        }
    }
    return ORC_OK;
}

}  // namespace orc
