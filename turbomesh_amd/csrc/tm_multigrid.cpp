#include "tm_multigrid.hpp"
#include "tm_smoother.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace tmh {

#define HIPCHK(x) hip_check((x), #x)

// log(g11 / g22) over ~4096 interior nodes of the caller's coordinates: its mean (-> the block's mean cell aspect ratio, which the
// semi-coarsening follows) and its standard deviation (what NO block-wide coarsening rule follows: TM_INNER_AUTO, tm_smoother.cpp)
static void log_aspect_stats(const double* xy, int ni, int nj, double& mean, double& sdev) {
    mean = sdev = 0.0;
    if (!xy || ni < 3 || nj < 3) return;
    const int si = std::max(1, (ni - 2) / 64), sj = std::max(1, (nj - 2) / 64);
    double sum = 0.0, sum2 = 0.0;
    long cnt = 0;
    for (int i = 1; i < ni - 1; i += si)
        for (int j = 1; j < nj - 1; j += sj) {
            const double* c = xy + 2 * (static_cast<size_t>(i) * nj + j);
            const double ax = c[2 * nj] - c[-2 * nj], ay = c[2 * nj + 1] - c[-2 * nj + 1];
            const double bx = c[2] - c[-2], by = c[3] - c[-1];
            const double g11 = ax * ax + ay * ay, g22 = bx * bx + by * by;
            if (g11 > 0.0 && g22 > 0.0) {
                const double l = std::log(g11 / g22);
                sum += l;
                sum2 += l * l;
                cnt += 1;
            }
        }
    if (!cnt) return;
    mean = sum / static_cast<double>(cnt);
    sdev = std::sqrt(std::max(0.0, sum2 / static_cast<double>(cnt) - mean * mean));
}

double BlockMG::aspect_of(const double* xy, int ni, int nj) {
    double mean, sdev;
    log_aspect_stats(xy, ni, nj, mean, sdev);
    return std::exp(mean);
}

double BlockMG::aspect_spread_of(const double* xy, int ni, int nj) {
    double mean, sdev;
    log_aspect_stats(xy, ni, nj, mean, sdev);
    return sdev;
}

void BlockMG::build(DeviceArena& arena, int ni, int nj, bool has_pq, double aspect, bool worst_case) {
    if (nj < 3 || ni < 3) throw TmError(TM_E_UNSUPPORTED, "multigrid: a block needs at least 3 x 3 nodes");
    if (const char* e = std::getenv("TM_MG_CYCLE")) {   // tuning knob: "nu_pre,nu_post,nu_coarsest,omega"
        int a = nu_pre, b = nu_post, c = nu_coarsest;
        double w = omega;
        if (std::sscanf(e, "%d,%d,%d,%lf", &a, &b, &c, &w) >= 2 && a >= 0 && b >= 0 && a + b >= 1 && c >= 0 && w > 0.0 && w < 2.0) {
            nu_pre = a;
            nu_post = b;
            nu_coarsest = c;
            omega = w;
        }
    }
    if (const char* e = std::getenv("TM_MG_FUSE_PROLONG")) {   // 0 = never, 1 = wherever it pays (default), 2 = on every level (tests)
        fuse_prolong = std::atoi(e) != 0;
        fuse_prolong_min = std::atoi(e) == 2 ? 0 : fuse_prolong_min;
    }
    if (const char* e = std::getenv("TM_MG_PAIR")) use_pair = std::atoi(e) != 0;   // 0: the one-sweep-per-pass kernels everywhere (A/B runs, bit-identity tests)
    if (const char* e = std::getenv("TM_MG_RESTRICT_FUSED")) fuse_restrict = std::atoi(e) != 0;
    L.clear();
    MgLevel l0;
    l0.ni = ni;
    l0.nj = nj;
    L.push_back(l0);
    if (worst_case) {   // tm_smoother_workspace_bytes: semi-coarsening levels sum to < 1 fine block per array, whatever the order
        const uint64_t n = static_cast<uint64_t>(ni) * nj + 64 * 256;
        for (int k = 0; k < (has_pq ? 6 : 5); ++k) (void)arena.alloc_n<double2>(n);
        return;
    }
    double ratio = aspect > 0.0 ? aspect : 1.0;   // g11/g22: > 1 = the j direction is the strongly coupled one
    while (L.size() < 24) {
        const MgLevel& f = L.back();
        MgLevel c;
        c.ci = f.ni >= 5;
        c.cj = f.nj >= 5;
        if (c.ci && c.cj) {
            if (ratio > 4.0) c.ci = 0;          // coarsen j only: g22 x4
            else if (ratio < 0.25) c.cj = 0;    // coarsen i only: g11 x4
        }
        if (!c.ci && !c.cj) break;
        if (c.ci && !c.cj) ratio *= 4.0;
        if (c.cj && !c.ci) ratio *= 0.25;
        c.ni = c.ci ? f.ni / 2 + 1 : f.ni;
        c.nj = c.cj ? f.nj / 2 + 1 : f.nj;
        const uint64_t n = static_cast<uint64_t>(c.ni) * c.nj;
        c.X = arena.alloc_n<double2>(n);
        if (has_pq) c.PQ = arena.alloc_n<double2>(n);
        c.f = arena.alloc_n<double2>(n);
        c.a = arena.alloc_n<double2>(n);
        c.b = arena.alloc_n<double2>(n);
        c.r = arena.alloc_n<double2>(n);
        for (double2* q : {c.f, c.a, c.b, c.r}) HIPCHK(hipMemset(q, 0, sizeof(double2) * n));   // perimeters stay zero for the lifetime of the handle
        L.push_back(c);
    }
}

MgPair BlockMG::pair(size_t fine) const {
    const MgLevel &f = L[fine], &c = L[fine + 1];
    MgPair g;
    g.nif = f.ni;
    g.njf = f.nj;
    g.nic = c.ni;
    g.njc = c.nj;
    g.ci = c.ci;
    g.cj = c.cj;
    return g;
}

void BlockMG::set_field(const double2* X0, const double2* PQ0, hipStream_t st) {
    L[0].X = const_cast<double2*>(X0);
    L[0].PQ = const_cast<double2*>(PQ0);
    for (size_t l = 0; l + 1 < L.size(); ++l) {
        const MgPair g = pair(l);
        HIPCHK(launch_mg_inject(L[l].X, L[l + 1].X, g, 1.0, 1.0, st));
        if (PQ0) HIPCHK(launch_mg_inject(L[l].PQ, L[l + 1].PQ, g, g.ci ? 2.0 : 1.0, g.cj ? 2.0 : 1.0, st));
    }
}

void BlockMG::smooth(const MgLevel& l, const double2* f, const double2* in, double2* out, hipStream_t st) const {
    ApplyBlock a;
    a.in = in;
    a.xk = l.X;
    a.pq = l.PQ;
    a.aux = f;
    a.out = out;
    a.ni = l.ni;
    a.nj = l.nj;
    a.omega = omega;
    a.partials = nullptr;
    HIPCHK(launch_apply_block(a, MODE_MG_SMOOTH, DOT_NONE, st));
}

void BlockMG::residual(const MgLevel& l, const double2* f, const double2* in, double2* out, hipStream_t st) const {
    ApplyBlock a;
    a.in = in;
    a.xk = l.X;
    a.pq = l.PQ;
    a.aux = f;
    a.out = out;
    a.ni = l.ni;
    a.nj = l.nj;
    a.omega = 0.0;
    a.partials = nullptr;
    HIPCHK(launch_apply_block(a, MODE_MG_RESID, DOT_NONE, st));
}

void BlockMG::vcycle(const double2* f0, double2* z, double2* w0, double2* w1, hipStream_t st) {
    const size_t nl = L.size();
    std::vector<double2*> cur(nl, nullptr), oth(nl, nullptr);
    std::vector<const double2*> rhs(nl, nullptr);
    // fine level: the last post-sweep must land in z; count the ping-pongs after the initial scaling
    const int pre_extra = nu_pre > 0 ? nu_pre - 1 : 0;
    // pair(l): the level runs its two pre-sweeps + residual, and its two post-sweeps (+ prolongation), as ONE pass each (k_mg_pair)
    auto pair_ok = [&](size_t l) { return use_pair && mg_pair_supported(L[l].ni, L[l].nj); };
    const bool pair_pre0 = pair_ok(0) && nu_pre == 2 && nl > 1, pair_post0 = pair_ok(0) && nu_post == 2;
    (void)pair_pre0;   // one flip either way
    const int flips0 = (nl == 1) ? pre_extra + nu_coarsest : pre_extra + (pair_post0 ? 1 : nu_post);
    cur[0] = (flips0 % 2 == 0) ? z : w0;
    oth[0] = (flips0 % 2 == 0) ? w0 : z;
    rhs[0] = f0;
    L[0].r = w1;
    for (size_t l = 1; l < nl; ++l) {
        cur[l] = L[l].a;
        oth[l] = L[l].b;
        rhs[l] = L[l].f;
    }
    auto sweeps = [&](size_t l, int n) {
        for (int k = 0; k < n; ++k) {
            smooth(L[l], rhs[l], cur[l], oth[l], st);
            std::swap(cur[l], oth[l]);
        }
    };
    // ---- down
    for (size_t l = 0; l < nl; ++l) {
        if (l == 0) {   // the two fine iterates are the caller's: make their perimeters zero (the coarse ones stay zero for good)
            HIPCHK(launch_copy_perimeter(w1, cur[0], L[0].ni, L[0].nj, st));
            HIPCHK(launch_copy_perimeter(w1, oth[0], L[0].ni, L[0].nj, st));
        }
        if (nu_pre == 2 && l + 1 < nl && pair_ok(l)) {   // both pre-sweeps AND the residual behind them in one pass over f
            MgPairArgs a;
            a.in = rhs[l];
            a.xk = L[l].X;
            a.pq = L[l].PQ;
            a.out = oth[l];
            a.out2 = L[l].r;
            a.ni = L[l].ni;
            a.nj = L[l].nj;
            a.omega = omega;
            const MgPair g = pair(l);
            if (fuse_restrict && mg_pair_restrict_supported(a.ni, a.nj, g.ci, g.cj)) {   // ... and the restriction behind the residual
                a.out2 = nullptr;
                a.xc = L[l + 1].X;
                a.fc = L[l + 1].f;
                a.nic = g.nic;
                a.njc = g.njc;
                a.ci = g.ci;
                a.cj = g.cj;
                HIPCHK(launch_mg_pair(a, 1, st));
                std::swap(cur[l], oth[l]);
                continue;
            }
            HIPCHK(launch_mg_pair(a, 1, st));
            std::swap(cur[l], oth[l]);
            HIPCHK(launch_mg_restrict(L[l].r, L[l + 1].X, L[l + 1].f, g, st));
            continue;
        }
        if (nu_pre >= 2) {   // sweeps 1 and 2 from e = 0 in one pass over f (K2, MODE_MG_FIRST2); it lands where scale + one sweep would
            ApplyBlock a;
            a.in = rhs[l];
            a.xk = L[l].X;
            a.pq = L[l].PQ;
            a.aux = nullptr;
            a.out = oth[l];
            a.ni = L[l].ni;
            a.nj = L[l].nj;
            a.omega = omega;
            a.partials = nullptr;
            HIPCHK(launch_apply_block(a, MODE_MG_FIRST2, DOT_NONE, st));
            std::swap(cur[l], oth[l]);
            sweeps(l, pre_extra - 1);
        } else {
            HIPCHK(launch_mg_scale(rhs[l], cur[l], L[l].ni, L[l].nj, nu_pre > 0 ? omega : 0.0, st));   // first sweep from e = 0
            sweeps(l, pre_extra);
        }
        if (l + 1 == nl) {
            sweeps(l, nu_coarsest);
            break;
        }
        residual(L[l], rhs[l], cur[l], L[l].r, st);
        HIPCHK(launch_mg_restrict(L[l].r, L[l + 1].X, L[l + 1].f, pair(l), st));
    }
    // ---- up
    for (size_t l = nl - 1; l-- > 0;) {
        if (nu_post == 2 && pair_ok(l)) {   // prolongation + both post-sweeps in one pass
            MgPairArgs a;
            a.in = cur[l];
            a.xk = L[l].X;
            a.pq = L[l].PQ;
            a.f = rhs[l];
            a.out = oth[l];
            a.coarse = cur[l + 1];
            const MgPair g = pair(l);
            a.nic = g.nic;
            a.njc = g.njc;
            a.ci = g.ci;
            a.cj = g.cj;
            a.ni = L[l].ni;
            a.nj = L[l].nj;
            a.omega = omega;
            HIPCHK(launch_mg_pair(a, 0, st));
            std::swap(cur[l], oth[l]);
            continue;
        }
        // (levels of a few hundred thousand nodes and more: below that both forms are launch-bound and the plain pair is as fast)
        if (fuse_prolong && nu_post >= 1 && static_cast<int64_t>(L[l].ni) * L[l].nj >= fuse_prolong_min) {
            // the correction is interpolated as the rows of the iterate enter the first post-smoothing sweep's window (the same
            // expression, the same bits as k_mg_prolong_add); the prolonged iterate itself is never stored
            ApplyBlock a;
            a.in = cur[l];
            a.in2 = cur[l + 1];
            a.xk = L[l].X;
            a.pq = L[l].PQ;
            a.aux = rhs[l];
            a.out = oth[l];
            a.ni = L[l].ni;
            a.nj = L[l].nj;
            a.omega = omega;
            a.partials = nullptr;
            HIPCHK(launch_mg_prolong_smooth(a, pair(l), st));
            std::swap(cur[l], oth[l]);
            sweeps(l, nu_post - 1);
        } else {
            HIPCHK(launch_mg_prolong_add(cur[l + 1], cur[l], pair(l), st));
            sweeps(l, nu_post);
        }
    }
    if (cur[0] != z) throw TmError(TM_E_ARG, "internal: multigrid result landed in the scratch block");
}

}  // namespace tmh
