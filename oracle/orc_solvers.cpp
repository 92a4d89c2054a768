// ORACLE (test infrastructure) -- linear solvers on the assembled CSR system.
//   bicgstab() : reference src/core/smoothing/BiCGStab.zig:155-448 (faithful)
//   gmres()    : reference src/core/smoothing/GMRES.zig:176-536   (faithful)
//   ILU(0)     : BiCGStab.zig:178-277, 384-422 (identical copy in GMRES.zig:199-298, 437-475)
//   banded_direct(): LU with partial pivoting -- stands in for the reference's UMFPACK
//                    backend (umfpack.zig:18-55): "the exact Picard iterate".
//   scaled_bicgstab(): BUILD-DEFINED (not in the reference): BiCGStab on the row-equilibrated
//                    system D^-1 A x = D^-1 b with a relative tolerance on the scaled residual
//                    (SURVEY.md H2).  Mirrors the recurrences the HIP path uses.
#include "orc_system.hpp"
#include <algorithm>
#include <cstring>

namespace orc {

namespace {

Float dot(const Float* a, const Float* b, Index n) {   // BiCGStab.zig:438-444
    Float sum = 0.0;
    for (Index i = 0; i < n; ++i) sum += a[i] * b[i];
    return sum;
}
Float norm(const Float* a, Index n) { return std::sqrt(dot(a, a, n)); }   // BiCGStab.zig:446-448

void matVec(const CsrView& A, const Float* x, Float* out) {   // BiCGStab.zig:424-435
    for (Index row = 0; row < A.n; ++row) {
        Float sum = 0.0;
        for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k) sum += A.v[k] * x[A.i[k]];
        out[row] = sum;
    }
}

struct Preconditioner {
    Precond kind;
    std::vector<Float> diag_inv;
    std::vector<Float> lu;
    std::vector<int32_t> diag_pos, marker;

    void updateDiagonalInverse(const CsrView& A) {   // BiCGStab.zig:155-175
        diag_inv.resize(A.n);
        for (Index row = 0; row < A.n; ++row) {
            Float diag = 0.0;
            for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k)
                if (A.i[k] == static_cast<int32_t>(row)) {
                    diag = A.v[k];
                    break;
                }
            diag_inv[row] = (diag == 0.0) ? 1.0 : 1.0 / diag;
        }
    }

    void updateIlu0(const CsrView& A) {   // BiCGStab.zig:178-277
        const Index nnz = static_cast<Index>(A.p[A.n]);
        lu.assign(A.v, A.v + nnz);
        diag_pos.assign(A.n, -1);
        marker.assign(A.n, -1);
        for (Index row = 0; row < A.n; ++row)
            for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k)
                if (A.i[k] == static_cast<int32_t>(row)) {
                    diag_pos[row] = k;
                    break;
                }
        for (Index row = 0; row < A.n; ++row) {
            const int32_t start = A.p[row], end = A.p[row + 1];
            for (int32_t k = start; k < end; ++k) marker[A.i[k]] = k;
            for (int32_t k = start; k < end; ++k) {
                const Index col = static_cast<Index>(A.i[k]);
                if (col >= row) continue;
                const int32_t diag_idx = diag_pos[col];
                Float diag = 1.0;
                if (diag_idx >= 0) {
                    diag = lu[diag_idx];
                    if (diag == 0.0) diag = 1.0;
                }
                const Float lij = lu[k] / diag;
                lu[k] = lij;
                for (int32_t r = A.p[col]; r < A.p[col + 1]; ++r) {
                    const Index col_k = static_cast<Index>(A.i[r]);
                    if (col_k <= col) continue;
                    const int32_t pos = marker[col_k];
                    if (pos >= 0) lu[pos] -= lij * lu[r];
                }
            }
            for (int32_t k = start; k < end; ++k) marker[A.i[k]] = -1;
        }
    }

    void update(const CsrView& A) {   // BiCGStab.zig:64-69
        if (kind == diagonal) updateDiagonalInverse(A);
        else updateIlu0(A);
    }

    void applyIlu0(const CsrView& A, const Float* rhs, Float* out) const {   // BiCGStab.zig:384-422
        for (Index row = 0; row < A.n; ++row) {
            Float sum = rhs[row];
            for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k) {
                const Index col = static_cast<Index>(A.i[k]);
                if (col < row) sum -= lu[k] * out[col];
            }
            out[row] = sum;
        }
        Index row = A.n;
        while (row > 0) {
            row -= 1;
            Float sum = out[row];
            for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k) {
                const Index col = static_cast<Index>(A.i[k]);
                if (col > row) sum -= lu[k] * out[col];
            }
            Float diag = 1.0;
            if (diag_pos[row] >= 0) {
                diag = lu[diag_pos[row]];
                if (diag == 0.0) diag = 1.0;
            }
            out[row] = sum / diag;
        }
    }

    void apply(const CsrView& A, const Float* rhs, Float* out) const {   // BiCGStab.zig:372-381
        if (kind == diagonal) {
            for (Index i = 0; i < A.n; ++i) out[i] = rhs[i] * diag_inv[i];
        } else {
            applyIlu0(A, rhs, out);
        }
    }
};

}  // namespace

// ILU(0) alone (BiCGStab.zig:178-277 factorisation, :384-422 application): the factor in the CSR's own pattern and M^-1 rhs, for tests that
// check a device implementation of the preconditioner bit for bit
void ilu0_factor_apply(const CsrView& A, const Float* rhs, Float* lu_out, Float* out) {
    Preconditioner M;
    M.kind = ilu0;
    M.update(A);
    const Index nnz = static_cast<Index>(A.p[A.n]);
    if (lu_out) std::copy(M.lu.begin(), M.lu.begin() + nnz, lu_out);
    if (rhs && out) M.applyIlu0(A, rhs, out);
}

// ---------------------------------------------------------------- BiCGStab.zig:279-370
SolveReport bicgstab(const CsrView& A, const Float* rhs, Float* x, Precond pc, Index max_iters, Float rtol, Float atol) {
    const Index n = A.n;
    const Float breakdown_eps = 1e-30;
    Preconditioner M;
    M.kind = pc;
    M.update(A);
    std::vector<Float> r(n), r_hat(n), p(n), v(n), s(n), t(n), precond(n);
    SolveReport rep;

    matVec(A, x, v.data());
    for (Index i = 0; i < n; ++i) r[i] = rhs[i] - v[i];
    r_hat = r;
    const Float norm_b = norm(rhs, n);
    Float norm_r = norm(r.data(), n);
    const Float tol = std::max(atol, rtol * norm_b);
    if (norm_r <= tol) return rep;
    std::fill(p.begin(), p.end(), 0.0);
    std::fill(v.begin(), v.end(), 0.0);
    Float rho_old = 1.0, alpha = 1.0, omega = 1.0;
    Index iter = 0;
    for (; iter < max_iters; ++iter) {
        const Float rho_new = dot(r_hat.data(), r.data(), n);
        if (std::fabs(rho_new) < breakdown_eps) break;
        const Float beta = (rho_new / rho_old) * (alpha / omega);
        for (Index i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * v[i]);
        M.apply(A, p.data(), precond.data());
        matVec(A, precond.data(), v.data());
        const Float denom = dot(r_hat.data(), v.data(), n);
        if (std::fabs(denom) < breakdown_eps) break;
        alpha = rho_new / denom;
        for (Index i = 0; i < n; ++i) s[i] = r[i] - alpha * v[i];
        for (Index i = 0; i < n; ++i) x[i] += alpha * precond[i];
        const Float norm_s = norm(s.data(), n);
        if (norm_s <= tol) {
            rep.iters = iter + 1;
            return rep;
        }
        M.apply(A, s.data(), precond.data());
        matVec(A, precond.data(), t.data());
        const Float t_dot_t = dot(t.data(), t.data(), n);
        if (std::fabs(t_dot_t) < breakdown_eps) break;
        omega = dot(t.data(), s.data(), n) / t_dot_t;
        if (std::fabs(omega) < breakdown_eps) break;
        for (Index i = 0; i < n; ++i) x[i] += omega * precond[i];
        for (Index i = 0; i < n; ++i) r[i] = s[i] - omega * t[i];
        norm_r = norm(r.data(), n);
        if (norm_r <= tol) {
            rep.iters = iter + 1;
            return rep;
        }
        rho_old = rho_new;
    }
    rep.iters = iter;
    rep.converged = false;   // BiCGStab.zig:368-369: warning only
    return rep;
}

// ---------------------------------------------------------------- GMRES.zig:510-524
namespace {
struct Givens {
    Float c, s, r;
};
Givens computeGivens(Float a, Float b) {
    if (b == 0.0) return {1.0, 0.0, a};
    if (std::fabs(b) > std::fabs(a)) {
        const Float t = a / b;
        const Float s = 1.0 / std::sqrt(1.0 + t * t);
        return {s * t, s, b / s};
    }
    const Float t = b / a;
    const Float c = 1.0 / std::sqrt(1.0 + t * t);
    return {c, c * t, a / c};
}
}  // namespace

// ---------------------------------------------------------------- GMRES.zig:300-423
SolveReport gmres(const CsrView& A, const Float* rhs, Float* x, Precond pc, Index restart_in, Index max_iters, Float rtol,
                  Float atol) {
    const Index n = A.n;
    const Float breakdown_eps = 1e-30;
    const Index restart = std::min(restart_in, n);   // GMRES.zig:93
    SolveReport rep;
    if (restart == 0) return rep;
    Preconditioner M;
    M.kind = pc;
    M.update(A);
    std::vector<Float> V((restart + 1) * n), H((restart + 1) * restart), cs(restart), sn(restart), g(restart + 1), r(n), w(n), z(n);
    auto hidx = [&](Index row, Index col) { return row + (restart + 1) * col; };   // GMRES.zig:495-497

    const Float norm_b = norm(rhs, n);
    const Float tol = std::max(atol, rtol * norm_b);
    Index iter_total = 0;
    while (iter_total < max_iters) {
        matVec(A, x, w.data());
        for (Index i = 0; i < n; ++i) r[i] = rhs[i] - w[i];
        M.apply(A, r.data(), z.data());
        const Float beta = norm(z.data(), n);
        if (beta <= tol) {
            rep.iters = iter_total;
            return rep;
        }
        for (Index i = 0; i < n; ++i) V[i] = z[i] / beta;
        std::fill(H.begin(), H.end(), 0.0);
        std::fill(cs.begin(), cs.end(), 0.0);
        std::fill(sn.begin(), sn.end(), 0.0);
        std::fill(g.begin(), g.end(), 0.0);
        g[0] = beta;
        Index cols_used = 0;
        bool converged = false;
        Float resid = beta;
        for (Index j = 0; j < restart && iter_total < max_iters; ++j) {
            matVec(A, &V[j * n], w.data());
            M.apply(A, w.data(), z.data());
            for (Index i = 0; i < j + 1; ++i) {
                const Float* vi = &V[i * n];
                const Float h_ij = dot(z.data(), vi, n);
                H[hidx(i, j)] = h_ij;
                for (Index k = 0; k < n; ++k) z[k] -= h_ij * vi[k];
            }
            const Float h_next = norm(z.data(), n);
            H[hidx(j + 1, j)] = h_next;
            if (h_next > breakdown_eps) {
                Float* vnext = &V[(j + 1) * n];
                for (Index k = 0; k < n; ++k) vnext[k] = z[k] / h_next;
            }
            for (Index i = 0; i < j; ++i) {
                const Float h_i = H[hidx(i, j)], h_ip1 = H[hidx(i + 1, j)];
                const Float temp = cs[i] * h_i + sn[i] * h_ip1;
                H[hidx(i + 1, j)] = -sn[i] * h_i + cs[i] * h_ip1;
                H[hidx(i, j)] = temp;
            }
            const Givens rot = computeGivens(H[hidx(j, j)], H[hidx(j + 1, j)]);
            cs[j] = rot.c;
            sn[j] = rot.s;
            H[hidx(j, j)] = rot.r;
            H[hidx(j + 1, j)] = 0.0;
            const Float g_j = g[j], g_jp1 = g[j + 1];
            g[j] = rot.c * g_j + rot.s * g_jp1;
            g[j + 1] = -rot.s * g_j + rot.c * g_jp1;
            resid = std::fabs(g[j + 1]);
            iter_total += 1;
            cols_used = j + 1;
            if (resid <= tol) {
                converged = true;
                break;
            }
        }
        if (cols_used == 0) break;
        Float* y = w.data();   // GMRES.zig:396
        Index idx = cols_used;
        while (idx > 0) {
            idx -= 1;
            Float sum = g[idx];
            for (Index k = idx + 1; k < cols_used; ++k) sum -= H[hidx(idx, k)] * y[k];
            const Float h_ii = H[hidx(idx, idx)];
            if (h_ii == 0.0) break;
            y[idx] = sum / h_ii;
        }
        for (Index i = 0; i < cols_used; ++i) {
            const Float* vi = &V[i * n];
            const Float yi = y[i];
            for (Index k = 0; k < n; ++k) x[k] += yi * vi[k];
        }
        if (converged || resid <= tol) {
            rep.iters = iter_total;
            return rep;
        }
    }
    rep.iters = iter_total;
    rep.converged = false;   // GMRES.zig:422: warning only
    return rep;
}

// ---------------------------------------------------------------- build-defined
// BiCGStab without preconditioner on the row-equilibrated system  (D^-1 A) x = D^-1 b,
// stopping on ||D^-1 (b - A x)||_2 <= max(atol, rtol * ||D^-1 b||_2).
SolveReport scaled_bicgstab(const CsrView& A, const Float* rhs, Float* x, Index max_iters, Float rtol, Float atol) {
    const Index n = A.n;
    const Float breakdown_eps = 1e-300;
    Preconditioner M;
    M.kind = diagonal;
    M.updateDiagonalInverse(A);
    const std::vector<Float>& dinv = M.diag_inv;
    auto applyA = [&](const Float* in, Float* out) {
        matVec(A, in, out);
        for (Index i = 0; i < n; ++i) out[i] *= dinv[i];
    };
    std::vector<Float> r(n), r_hat(n), p(n, 0.0), v(n, 0.0), s(n), t(n), bs(n);
    SolveReport rep;
    for (Index i = 0; i < n; ++i) bs[i] = rhs[i] * dinv[i];
    applyA(x, v.data());
    for (Index i = 0; i < n; ++i) r[i] = bs[i] - v[i];
    r_hat = r;
    const Float tol = std::max(atol, rtol * norm(bs.data(), n));
    if (norm(r.data(), n) <= tol) return rep;
    std::fill(v.begin(), v.end(), 0.0);
    Float rho_old = 1.0, alpha = 1.0, omega = 1.0;
    Index iter = 0;
    for (; iter < max_iters; ++iter) {
        const Float rho_new = dot(r_hat.data(), r.data(), n);
        if (std::fabs(rho_new) < breakdown_eps) break;
        const Float beta = (rho_new / rho_old) * (alpha / omega);
        for (Index i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * v[i]);
        applyA(p.data(), v.data());
        const Float denom = dot(r_hat.data(), v.data(), n);
        if (std::fabs(denom) < breakdown_eps) break;
        alpha = rho_new / denom;
        for (Index i = 0; i < n; ++i) s[i] = r[i] - alpha * v[i];
        for (Index i = 0; i < n; ++i) x[i] += alpha * p[i];
        if (norm(s.data(), n) <= tol) {
            rep.iters = iter + 1;
            return rep;
        }
        applyA(s.data(), t.data());
        const Float tt = dot(t.data(), t.data(), n);
        if (std::fabs(tt) < breakdown_eps) break;
        omega = dot(t.data(), s.data(), n) / tt;
        for (Index i = 0; i < n; ++i) x[i] += omega * s[i];
        for (Index i = 0; i < n; ++i) r[i] = s[i] - omega * t[i];
        if (norm(r.data(), n) <= tol) {
            rep.iters = iter + 1;
            return rep;
        }
        if (std::fabs(omega) < breakdown_eps) break;
        rho_old = rho_new;
    }
    rep.iters = iter;
    rep.converged = false;
    return rep;
}

// ---------------------------------------------------------------- stands in for umfpack.zig:29-55
// Banded LU with partial pivoting (LAPACK dgbtf2-style storage), exact up to rounding.
void banded_direct(const CsrView& A, const Float* rhs, Float* x) {
    const Index n = A.n;
    Index kl = 0, ku = 0;
    for (Index row = 0; row < n; ++row)
        for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k) {
            const Index col = static_cast<Index>(A.i[k]);
            if (col < row) kl = std::max(kl, row - col);
            else ku = std::max(ku, col - row);
        }
    const Index ldab = 2 * kl + ku + 1;
    if (static_cast<double>(ldab) * static_cast<double>(n) * 8.0 > 8e9)
        throw Error(ORC_E_MEMORY, "banded direct solve would need more than 8 GB (bandwidth " + std::to_string(kl + ku) + ")");
    std::vector<Float> ab(ldab * n, 0.0);   // column-major band storage: A(i,j) at ab[(kl+ku+i-j) + j*ldab]
    auto AB = [&](Index i, Index j) -> Float& { return ab[(kl + ku + i - j) + j * ldab]; };
    for (Index row = 0; row < n; ++row)
        for (int32_t k = A.p[row]; k < A.p[row + 1]; ++k) AB(row, static_cast<Index>(A.i[k])) += A.v[k];
    std::vector<Float> b(rhs, rhs + n);
    std::vector<Index> piv(n);
    for (Index j = 0; j < n; ++j) {
        const Index last = std::min(n - 1, j + kl);
        Index p = j;
        Float best = std::fabs(AB(j, j));
        for (Index i = j + 1; i <= last; ++i)
            if (std::fabs(AB(i, j)) > best) {
                best = std::fabs(AB(i, j));
                p = i;
            }
        if (best == 0.0) throw Error(ORC_E_SINGULAR, "singular matrix in banded direct solve");
        piv[j] = p;
        const Index cmax = std::min(n - 1, j + ku + kl);
        if (p != j) {
            for (Index c = j; c <= cmax; ++c) std::swap(AB(j, c), AB(p, c));
            std::swap(b[j], b[p]);
        }
        const Float d = AB(j, j);
        for (Index i = j + 1; i <= last; ++i) {
            const Float l = AB(i, j) / d;
            if (l == 0.0) continue;
            AB(i, j) = l;
            for (Index c = j + 1; c <= cmax; ++c) AB(i, c) -= l * AB(j, c);
            b[i] -= l * b[j];
        }
    }
    for (Index jj = n; jj > 0; --jj) {
        const Index j = jj - 1;
        Float sum = b[j];
        const Index cmax = std::min(n - 1, j + ku + kl);
        for (Index c = j + 1; c <= cmax; ++c) sum -= AB(j, c) * x[c];
        x[j] = sum / AB(j, j);
    }
}

}  // namespace orc
