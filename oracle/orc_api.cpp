// ORACLE (test infrastructure) -- extern "C" surface + the smooth.mesh outer loop
// (reference src/core/smoothing/smooth.zig:74-166, solver dispatch solver.zig:40-93).
#include "orc_system.hpp"
#include "tm_oracle.h"
#include <chrono>
#include <cstring>
#include <memory>

namespace orc {
void cluster_uniform(Float*, Index);
void cluster_roberts(Float*, Index, Float, Float);
void cluster_tanh(Float*, Index, Float);
void line_interpolate(Vec2d, Vec2d, const Float*, Index, Vec2d*);
Index edge_combine_len(Index, const uint64_t*, const uint64_t*);
int edge_combine(Index, const Vec2d* const*, const Float* const*, const uint64_t*, const uint64_t*, Vec2d*, Float*);
int tfi_block(Vec2d*, Index, Index, const Vec2d*, const Vec2d*, const Vec2d*, const Vec2d*, const Float*, const Float*,
              const Float*, const Float*);
int tfi_linear2d(Vec2d*, Index, Index, const Vec2d*, const Vec2d*, const Vec2d*, const Vec2d*);
}  // namespace orc

using namespace orc;

static thread_local std::string g_last_error;
const char* orc_last_error(void) { return g_last_error.c_str(); }

template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of memory";
        return ORC_E_MEMORY;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return ORC_E_ARG;
    }
}

static Mesh meshFromDesc(const orc_mesh_desc* d) {
    if (!d || (d->nblocks && !d->blocks)) throw Error(ORC_E_ARG, "null mesh description");
    Mesh m;
    for (uint64_t b = 0; b < d->nblocks; ++b) {
        Block blk;
        blk.ni = d->blocks[b].ni;
        blk.nj = d->blocks[b].nj;
        blk.pts = reinterpret_cast<Vec2d*>(d->blocks[b].xy);
        if (!blk.pts) throw Error(ORC_E_ARG, "null block coordinates");
        m.blocks.push_back(blk);
    }
    auto toRange = [&](const orc_range& r) {
        if (r.block >= d->nblocks || r.side > 3) throw Error(ORC_E_TOPOLOGY, "range refers to a missing block/side");
        const Block& b = m.blocks[r.block];
        const Index lim = (r.side == ORC_SIDE_I_MIN || r.side == ORC_SIDE_I_MAX) ? b.ni : b.nj;
        if (r.start >= lim || r.end >= lim) throw Error(ORC_E_TOPOLOGY, "range exceeds the block side");
        return Range{static_cast<Index>(r.block), static_cast<Side>(r.side), static_cast<Index>(r.start), static_cast<Index>(r.end)};
    };
    for (uint64_t c = 0; c < d->nconns; ++c) {
        Connection conn;
        conn.ranges[0] = toRange(d->conns[c].r[0]);
        conn.ranges[1] = toRange(d->conns[c].r[1]);
        conn.has_periodicity = d->conns[c].has_periodicity != 0;
        conn.periodicity = vinit(d->conns[c].periodicity[0], d->conns[c].periodicity[1]);
        m.connections.push_back(conn);
    }
    for (uint64_t c = 0; c < d->nbcs; ++c) {
        if (d->bcs[c].kind > 2) throw Error(ORC_E_TOPOLOGY, "unknown boundary condition kind");
        m.boundary_conditions.push_back(Condition{toRange(d->bcs[c].range), static_cast<ConditionTag>(d->bcs[c].kind)});
    }
    return m;
}

struct orc_system {
    System sys;
};

// One Solver.solve(): fillXSpecific, solve x, fillYSpecific, solve y  (BiCGStab.zig:71-85,
// GMRES.zig:76-90, umfpack.zig:18-24).
static void solveOnce(System& s, const orc_solver_opt& opt, uint64_t* inner, int32_t* notconv) {
    const bool iterative = opt.tag != ORC_SOLVER_DIRECT;
    if (iterative && !s.seeded_initial_guess) s.seedInitialGuess();
    const CsrView A{s.dof, s.lhs_p.data(), s.lhs_i.data(), s.lhs_values.data()};
    auto one = [&](const Float* rhs, Float* x) {
        SolveReport rep;
        switch (opt.tag) {
            case ORC_SOLVER_BICGSTAB:
                rep = bicgstab(A, rhs, x, static_cast<Precond>(opt.preconditioner), 1000, 1e-6, 1e-8);   // BiCGStab.zig:19-21
                break;
            case ORC_SOLVER_GMRES:
                rep = gmres(A, rhs, x, static_cast<Precond>(opt.preconditioner), 30, 1000, 1e-6, 1e-8);   // GMRES.zig:21-24
                break;
            case ORC_SOLVER_DIRECT: banded_direct(A, rhs, x); break;
            case ORC_SOLVER_SCALED_BICGSTAB:
                rep = scaled_bicgstab(A, rhs, x, opt.max_iters ? opt.max_iters : 100000, opt.rtol > 0 ? opt.rtol : 1e-13, opt.atol);
                break;
            default: throw Error(ORC_E_ARG, "unknown solver tag (ExternalSolverNotEnabled, solver.zig:48)");
        }
        if (inner) *inner += rep.iters;
        if (notconv && !rep.converged) *notconv += 1;
    };
    s.fillXSpecific();
    one(s.rhs_x.data(), s.x_new.data());
    s.fillYSpecific();
    one(s.rhs_y.data(), s.y_new.data());
}

extern "C" {

void orc_cluster_uniform(double* u, uint64_t n) { cluster_uniform(u, n); }
void orc_cluster_roberts(double* u, uint64_t n, double a, double b) { cluster_roberts(u, n, a, b); }
void orc_cluster_tanh(double* u, uint64_t n, double ds) { cluster_tanh(u, n, ds); }

uint64_t orc_edge_combine_len(uint64_t nviews, const uint64_t* start, const uint64_t* end) {
    return edge_combine_len(nviews, start, end);
}
int orc_edge_combine(uint64_t nviews, const double* const* points, const double* const* clus, const uint64_t* start,
                     const uint64_t* end, double* out_points, double* out_clus) {
    return edge_combine(nviews, reinterpret_cast<const Vec2d* const*>(points), clus, start, end,
                        reinterpret_cast<Vec2d*>(out_points), out_clus);
}
void orc_line_interpolate(const double start[2], const double end[2], const double* u, uint64_t n, double* out_xy) {
    line_interpolate(vinit(start[0], start[1]), vinit(end[0], end[1]), u, n, reinterpret_cast<Vec2d*>(out_xy));
}

int orc_tfi_block(double* xy_out, uint64_t ni, uint64_t nj, const double* a, const double* b, const double* c, const double* d,
                  const double* s1, const double* s2, const double* t1, const double* t2) {
    return tfi_block(reinterpret_cast<Vec2d*>(xy_out), ni, nj, reinterpret_cast<const Vec2d*>(a), reinterpret_cast<const Vec2d*>(b),
                     reinterpret_cast<const Vec2d*>(c), reinterpret_cast<const Vec2d*>(d), s1, s2, t1, t2);
}
int orc_tfi_linear2d(double* xy_out, uint64_t ni, uint64_t nj, const double* a, const double* b, const double* c, const double* d) {
    return tfi_linear2d(reinterpret_cast<Vec2d*>(xy_out), ni, nj, reinterpret_cast<const Vec2d*>(a), reinterpret_cast<const Vec2d*>(b),
                        reinterpret_cast<const Vec2d*>(c), reinterpret_cast<const Vec2d*>(d));
}

// smooth.zig:74-166
int orc_smooth_mesh(const orc_mesh_desc* mesh, uint64_t iterations, const orc_solver_opt* opt, const orc_control_fn* cf,
                    orc_stats* stats, double* residual_history) {
    return guarded([&]() {
        if (!opt) throw Error(ORC_E_ARG, "null solver option");
        System s;
        const int algo = cf ? cf->kind : ORC_CF_LAPLACE;
        s.init(meshFromDesc(mesh), algo, cf ? White{cf->ds_target, cf->theta_target} : White{0, 0});
        orc_stats st;
        std::memset(&st, 0, sizeof(st));
        for (uint64_t n = 0; n < iterations; ++n) {
            s.fill(n);
            solveOnce(s, *opt, &st.inner_iterations, &st.not_converged);
            st.last_residual = s.commit(&st.last_dx2, &st.last_dy2);
            if (residual_history) residual_history[n] = st.last_residual;
            st.outer_iterations += 1;
        }
        if (stats) *stats = st;
        return ORC_OK;
    });
}

orc_system* orc_system_create(const orc_mesh_desc* mesh, const orc_control_fn* cf, int* err) {
    orc_system* h = nullptr;
    const int rc = guarded([&]() {
        auto p = std::make_unique<orc_system>();
        const int algo = cf ? cf->kind : ORC_CF_LAPLACE;
        p->sys.init(meshFromDesc(mesh), algo, cf ? White{cf->ds_target, cf->theta_target} : White{0, 0});
        h = p.release();
        return ORC_OK;
    });
    if (err) *err = rc;
    return h;
}
void orc_system_destroy(orc_system* s) { delete s; }
int orc_system_fill(orc_system* s, uint64_t iteration) { return guarded([&]() { s->sys.fill(iteration); return ORC_OK; }); }
int orc_system_fill_x_specific(orc_system* s) { return guarded([&]() { s->sys.fillXSpecific(); return ORC_OK; }); }
int orc_system_fill_y_specific(orc_system* s) { return guarded([&]() { s->sys.fillYSpecific(); return ORC_OK; }); }
uint64_t orc_system_dof(const orc_system* s) { return s->sys.dof; }
uint64_t orc_system_nnz(const orc_system* s) { return s->sys.lhs_i.size(); }
const int32_t* orc_system_lhs_p(const orc_system* s) { return s->sys.lhs_p.data(); }
const int32_t* orc_system_lhs_i(const orc_system* s) { return s->sys.lhs_i.data(); }
double* orc_system_lhs_values(orc_system* s) { return s->sys.lhs_values.data(); }
double* orc_system_rhs_x(orc_system* s) { return s->sys.rhs_x.data(); }
double* orc_system_rhs_y(orc_system* s) { return s->sys.rhs_y.data(); }
double* orc_system_x_new(orc_system* s) { return s->sys.x_new.data(); }
double* orc_system_y_new(orc_system* s) { return s->sys.y_new.data(); }
double* orc_system_control_function(orc_system* s) { return reinterpret_cast<double*>(s->sys.control_function.data.data()); }
uint64_t orc_system_nboundary(const orc_system* s) { return s->sys.boundary_points.kind.size(); }
const int32_t* orc_system_boundary_kind(const orc_system* s) { return s->sys.boundary_points.kind.data(); }
void orc_system_seed_initial_guess(orc_system* s) { s->sys.seedInitialGuess(); }
double orc_system_commit(orc_system* s, double* dx2, double* dy2) { return s->sys.commit(dx2, dy2); }
int orc_system_solve(orc_system* s, const orc_solver_opt* opt, uint64_t* inner_iters, int32_t* not_converged) {
    return guarded([&]() {
        if (inner_iters) *inner_iters = 0;
        if (not_converged) *not_converged = 0;
        solveOnce(s->sys, *opt, inner_iters, not_converged);
        return ORC_OK;
    });
}
void orc_system_matvec(const orc_system* s, const double* x, double* out) { s->sys.matVec(x, out); }

int orc_csr_bicgstab(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* b, double* x, int precond,
                     uint64_t max_iters, double rtol, double atol, uint64_t* iters) {
    return guarded([&]() {
        const SolveReport r = bicgstab(CsrView{n, Ap, Ai, Ax}, b, x, static_cast<Precond>(precond), max_iters, rtol, atol);
        if (iters) *iters = r.iters;
        return r.converged ? ORC_OK : 1;
    });
}
int orc_csr_ilu0(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* rhs, double* lu_out, double* out) {
    return guarded([&]() {
        ilu0_factor_apply(CsrView{n, Ap, Ai, Ax}, rhs, lu_out, out);
        return ORC_OK;
    });
}
int orc_csr_gmres(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* b, double* x, int precond,
                  uint64_t restart, uint64_t max_iters, double rtol, double atol, uint64_t* iters) {
    return guarded([&]() {
        const SolveReport r = gmres(CsrView{n, Ap, Ai, Ax}, b, x, static_cast<Precond>(precond), restart, max_iters, rtol, atol);
        if (iters) *iters = r.iters;
        return r.converged ? ORC_OK : 1;
    });
}
int orc_csr_direct(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* b, double* x) {
    return guarded([&]() {
        banded_direct(CsrView{n, Ap, Ai, Ax}, b, x);
        return ORC_OK;
    });
}

// ---- cpu_baseline timing helpers (single thread, same arithmetic as above) ----
double orc_time_bicgstab_iterations(uint64_t ni, uint64_t nj, double* xy, uint64_t iters, double* fill_seconds) {
    using clk = std::chrono::steady_clock;
    orc_block blk{xy, ni, nj};
    orc_mesh_desc d{&blk, 1, nullptr, 0, nullptr, 0};
    System s;
    s.init(meshFromDesc(&d), ORC_CF_LAPLACE, White{0, 0});
    auto t0 = clk::now();
    s.fill(0);
    auto t1 = clk::now();
    if (fill_seconds) *fill_seconds = std::chrono::duration<double>(t1 - t0).count();
    s.seedInitialGuess();
    s.fillXSpecific();
    // perturb the warm start so the solver does not return before iterating (rtol/atol = 0
    // disables the stop test; exactly `iters` reference iterations are executed and timed)
    for (Index i = 0; i < s.dof; ++i) s.x_new[i] += 1e-3 * std::sin(static_cast<double>(i));
    const CsrView A{s.dof, s.lhs_p.data(), s.lhs_i.data(), s.lhs_values.data()};
    auto t2 = clk::now();
    bicgstab(A, s.rhs_x.data(), s.x_new.data(), diagonal, iters, 0.0, 0.0);
    auto t3 = clk::now();
    return std::chrono::duration<double>(t3 - t2).count();
}

// Stage timings of the reference's CPU path for ONE outer iteration on one block (single thread, the reference has no threads):
//   out[0] RowCompressedMatrixSystem2d.init (smooth.zig:309-385: pattern, row kinds, static rows)
//   out[1] system.fill(0) (smooth.zig:923-1113)
//   out[2] BiCGStab + diagonal preconditioner (BiCGStab.zig:279-370), out[3] = iterations executed (x-system)
//   out[4] ILU(0) factorisation + GMRES workspace (GMRES.zig:199-298; a solve with 0 iterations)
//   out[5] GMRES(30) + ILU(0) (GMRES.zig:300-423; factorisation included), out[6] = iterations executed (y-system)
//   out[7] residual + copy-back (smooth.zig:112-153)
// rtol = atol = 0 disables the stop tests, so exactly the requested iteration counts are executed and timed.
int orc_time_reference_path(uint64_t ni, uint64_t nj, double* xy, uint64_t bicg_iters, uint64_t gmres_iters, double* out) {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    try {
        orc_block blk{xy, ni, nj};
        orc_mesh_desc d{&blk, 1, nullptr, 0, nullptr, 0};
        System s;
        auto t0 = clk::now();
        s.init(meshFromDesc(&d), ORC_CF_LAPLACE, White{0, 0});
        auto t1 = clk::now();
        s.fill(0);
        auto t2 = clk::now();
        out[0] = secs(t0, t1);
        out[1] = secs(t1, t2);
        s.seedInitialGuess();
        for (Index i = 0; i < s.dof; ++i) {   // perturbed warm start: the solvers must not return before iterating
            s.x_new[i] += 1e-3 * std::sin(static_cast<double>(i));
            s.y_new[i] += 1e-3 * std::cos(static_cast<double>(i));
        }
        const CsrView A{s.dof, s.lhs_p.data(), s.lhs_i.data(), s.lhs_values.data()};
        s.fillXSpecific();
        auto t3 = clk::now();
        const SolveReport rb = bicgstab(A, s.rhs_x.data(), s.x_new.data(), diagonal, bicg_iters, 0.0, 0.0);
        auto t4 = clk::now();
        out[2] = secs(t3, t4);
        out[3] = static_cast<double>(rb.iters);
        s.fillYSpecific();
        auto t5 = clk::now();
        (void)gmres(A, s.rhs_y.data(), s.y_new.data(), ilu0, 30, 0, 0.0, 0.0);
        auto t6 = clk::now();
        const SolveReport rg = gmres(A, s.rhs_y.data(), s.y_new.data(), ilu0, 30, gmres_iters, 0.0, 0.0);
        auto t7 = clk::now();
        out[4] = secs(t5, t6);
        out[5] = secs(t6, t7);
        out[6] = static_cast<double>(rg.iters);
        Float dx2 = 0, dy2 = 0;
        auto t8 = clk::now();
        (void)s.commit(&dx2, &dy2);
        auto t9 = clk::now();
        out[7] = secs(t8, t9);
        return 0;
    } catch (const std::exception& e) {
        return -1;
    }
}

}  // extern "C"
