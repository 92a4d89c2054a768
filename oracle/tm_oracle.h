/*
 * tm_oracle.h -- C ABI of the CPU ORACLE (test infrastructure, NOT the product).
 *
 * The oracle is a plain CPU restatement of turbomesh's src/core hot path
 * (TFI seeding + Winslow/Poisson elliptic smoothing).  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check /
 * time the HIP path against the reference's arithmetic.  Nothing under
 * turbomesh_amd/ may include, link or call it.
 *
 * PARITY STATUS: "parity unpinned" for TFI and smoothing -- the reference's own
 * tests hold no golden vector for either.  The adjacent
 * known-answer vectors the reference does hold are checked in tests/:
 *   - Edge.combine exact vectors           (reference src/core/discrete.zig:219-290)
 *   - 5x5 sparse system x = (1,2,3,4,5)    (reference src/core/smoothing/umfpack.zig:71-97)
 *   - commented 3x3 TFI case               (reference src/core/tfi.zig:230-260)
 * The reference is Zig 0.15.2 and cannot be compiled in this image (no zig), so
 * there is no oracle/_ref build.
 *
 * POD layouts below are deliberately identical to include/tm_hip.h so a test can
 * describe a mesh once and hand it to both sides.
 */
#ifndef TM_ORACLE_H
#define TM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* boundary.zig:8-13 -- declaration order */
enum { ORC_SIDE_I_MIN = 0, ORC_SIDE_I_MAX = 1, ORC_SIDE_J_MIN = 2, ORC_SIDE_J_MAX = 3 };
/* boundary.zig:178-182 */
enum { ORC_BC_WALL = 0, ORC_BC_INLET = 1, ORC_BC_OUTLET = 2 };

typedef struct { uint64_t block; uint32_t side; uint32_t _pad; uint64_t start; uint64_t end; } orc_range;
typedef struct { orc_range r[2]; int32_t has_periodicity; int32_t _pad; double periodicity[2]; } orc_connection;
typedef struct { orc_range range; uint32_t kind; uint32_t _pad; } orc_condition;
typedef struct { double* xy; uint64_t ni; uint64_t nj; } orc_block;
typedef struct {
    orc_block* blocks; uint64_t nblocks;
    orc_connection* conns; uint64_t nconns;
    orc_condition* bcs; uint64_t nbcs;
} orc_mesh_desc;

/* wall_control_function.zig:10-20, 56-68 */
enum { ORC_CF_LAPLACE = 0, ORC_CF_WHITE = 1 };
typedef struct { int32_t kind; int32_t _pad; double ds_target; double theta_target; } orc_control_fn;

/* solver.zig:10-27 + build-defined oracle-only modes (documented in DESIGN.md) */
enum {
    ORC_SOLVER_GMRES = 0,      /* GMRES.zig, faithful                                  */
    ORC_SOLVER_BICGSTAB = 1,   /* BiCGStab.zig, faithful                               */
    ORC_SOLVER_DIRECT = 2,     /* banded LU, partial pivoting = umfpack.zig semantics  */
    ORC_SOLVER_SCALED_BICGSTAB = 4 /* build-defined: BiCGStab on D^-1 A, scale-aware tol */
};
enum { ORC_PRECOND_DIAGONAL = 0, ORC_PRECOND_ILU0 = 1 };
typedef struct {
    int32_t tag; int32_t preconditioner;
    /* only used by ORC_SOLVER_SCALED_BICGSTAB; the faithful solvers keep the
     * reference's hard-coded max_iters=1000, rtol=1e-6, atol=1e-8, restart=30 */
    double rtol; double atol; uint64_t max_iters;
} orc_solver_opt;

typedef struct {
    uint64_t outer_iterations;
    uint64_t inner_iterations;      /* total inner iterations over both components */
    double   last_residual;         /* (sum dx^2 + sum dy^2)^2, smooth.zig:136 */
    double   last_dx2, last_dy2;
    int32_t  not_converged;         /* number of inner solves that hit max_iters (warning only) */
    int32_t  _pad;
} orc_stats;

enum { ORC_OK = 0, ORC_E_SIZE = -1, ORC_E_TOPOLOGY = -2, ORC_E_MISMATCH = -3, ORC_E_OVERFLOW = -4,
       ORC_E_SINGULAR = -5, ORC_E_ARG = -6, ORC_E_MEMORY = -7 };

const char* orc_last_error(void);

/* ---- clustering.zig ---- */
void orc_cluster_uniform(double* u, uint64_t n);                                   /* :9-17  */
void orc_cluster_roberts(double* u, uint64_t n, double alpha, double beta);        /* :24-42 */
void orc_cluster_tanh(double* u, uint64_t n, double delta_s);                      /* :56-95 */

/* ---- discrete.zig:38-136  Edge.combine over nviews EdgeViews ----
 * points[v]: pointer to that view's edge points (x,y interleaved), clus[v]: its clustering,
 * start[v], end[v] inclusive (start > end = reversed).  out_points / out_clus sized by
 * orc_edge_combine_len.  Returns ORC_E_MISMATCH if consecutive end points differ by > 1e-10. */
uint64_t orc_edge_combine_len(uint64_t nviews, const uint64_t* start, const uint64_t* end);
int orc_edge_combine(uint64_t nviews, const double* const* points, const double* const* clus,
                     const uint64_t* start, const uint64_t* end, double* out_points, double* out_clus);

/* ---- geometry.zig:21-40 Line.interpolate ---- */
void orc_line_interpolate(const double start[2], const double end[2], const double* u, uint64_t n, double* out_xy);

/* ---- tfi.zig ---- */
int orc_tfi_block(double* xy_out, uint64_t ni, uint64_t nj,
                  const double* x_i_min, const double* x_i_max,
                  const double* x_j_min, const double* x_j_max,
                  const double* s1, const double* s2, const double* t1, const double* t2);   /* :112-208 */
int orc_tfi_linear2d(double* xy_out, uint64_t ni, uint64_t nj,
                     const double* e_i_min, const double* e_i_max,
                     const double* e_j_min, const double* e_j_max);                           /* :19-67 */

/* ---- smooth.zig: whole smoother (seam 1) ---- */
int orc_smooth_mesh(const orc_mesh_desc* mesh, uint64_t iterations, const orc_solver_opt* opt,
                    const orc_control_fn* cf, orc_stats* stats, double* residual_history /* [iterations] or NULL */);

/* ---- smooth.zig: stepping interface over RowCompressedMatrixSystem2d (seam 2) ----
 * Lets a test drive the Picard loop with an independent solver (scipy splu) on the
 * oracle-assembled CSR, i.e. the reference's UMFPACK semantics. */
typedef struct orc_system orc_system;
orc_system* orc_system_create(const orc_mesh_desc* mesh, const orc_control_fn* cf, int* err);
void orc_system_destroy(orc_system*);
int  orc_system_fill(orc_system*, uint64_t iteration);       /* smooth.zig:1107-1113 */
int  orc_system_fill_x_specific(orc_system*);                /* smooth.zig:1115-1143 */
int  orc_system_fill_y_specific(orc_system*);                /* smooth.zig:1145-1165 */
uint64_t orc_system_dof(const orc_system*);
uint64_t orc_system_nnz(const orc_system*);
const int32_t* orc_system_lhs_p(const orc_system*);
const int32_t* orc_system_lhs_i(const orc_system*);
double* orc_system_lhs_values(orc_system*);
double* orc_system_rhs_x(orc_system*);
double* orc_system_rhs_y(orc_system*);
double* orc_system_x_new(orc_system*);
double* orc_system_y_new(orc_system*);
double* orc_system_control_function(orc_system*);            /* (P,Q) interleaved, dof entries */
uint64_t orc_system_nboundary(const orc_system*);
const int32_t* orc_system_boundary_kind(const orc_system*);  /* BlockBoundaryPointKind per perimeter buffer index */
void orc_system_seed_initial_guess(orc_system*);             /* BiCGStab.zig:136-153 */
/* residual + copy-back (smooth.zig:112-153); returns (sx+sy)^2, optionally sx, sy */
double orc_system_commit(orc_system*, double* dx2, double* dy2);
/* one faithful Solver.solve() (x then y) with the given option */
int  orc_system_solve(orc_system*, const orc_solver_opt* opt, uint64_t* inner_iters, int32_t* not_converged);
/* CSR mat-vec with the currently filled values (BiCGStab.zig:424-435) */
void orc_system_matvec(const orc_system*, const double* x, double* out);

/* ---- stand-alone CSR solvers (for the umfpack.zig 5x5 KAT etc.) ---- */
int orc_csr_bicgstab(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* b, double* x,
                     int precond, uint64_t max_iters, double rtol, double atol, uint64_t* iters);
int orc_csr_gmres(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* b, double* x,
                  int precond, uint64_t restart, uint64_t max_iters, double rtol, double atol, uint64_t* iters);
int orc_csr_direct(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* b, double* x);
/* ILU(0) alone: the factor in the CSR's own pattern (BiCGStab.zig:178-277) into lu_out [nnz] and M^-1 rhs (:384-422) into out [n];
 * rhs / out may be NULL (factor only) */
int orc_csr_ilu0(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* rhs, double* lu_out, double* out);

/* ---- timing helpers for bench.py's cpu_baseline leg (single thread) ---- */
/* One reference-style inner BiCGStab(diagonal) iteration block on an ni x nj single block:
 * assembles the CSR once (fill), then runs `iters` BiCGStab iterations on the x system and
 * returns elapsed seconds of the iteration loop only (2 mat-vecs per iteration). */
double orc_time_bicgstab_iterations(uint64_t ni, uint64_t nj, double* xy, uint64_t iters, double* fill_seconds);
/* per-stage seconds of ONE outer iteration of the reference's CPU path on one block (see orc_api.cpp): out[8] */
int orc_time_reference_path(uint64_t ni, uint64_t nj, double* xy, uint64_t bicg_iters, uint64_t gmres_iters, double* out);
/* MIRROR of the device's interior-row evaluation (orc_mirror.cpp): mode 0 raw A w, 1 D^-1 A w, 2 -D^-1 A w,
 * 3 relax w + omega(-D^-1 A w); interior rows of one ni x nj block, perimeter of `out` untouched. */
int orc_mirror_apply_block(int mode, uint64_t ni, uint64_t nj, const double* in, const double* xk, const double* pq /* or NULL */,
                           double* out, double omega);
/* `sweeps` matrix-free Jacobi elliptic sweeps of a fixed-boundary block in the device operation order. */
double orc_time_relax_sweeps(uint64_t ni, uint64_t nj, double* xy, double* scratch, uint64_t sweeps, double omega);
/* the same with every sweep's rows split over `threads` host threads (not the reference's behaviour: it is single-threaded) */
double orc_time_relax_sweeps_mt(uint64_t ni, uint64_t nj, double* xy, double* scratch, uint64_t sweeps, double omega, uint32_t threads);

/* the reference's libm for the White control function (orc_refmath.hpp: Zig std.math = musl): out_acos[i] = acos(x[i]),
 * out_atan2[i] = atan2(y[i], x[i]) */
void orc_ref_white_math(const double* x, const double* y, uint64_t n, double* out_acos, double* out_atan2);

#ifdef __cplusplus
}
#endif
#endif
