/*
 * tm_hip.h -- C ABI of libtm_hip.so: MI355X (gfx950) structured-grid smoother for
 * turbomesh's src/core hot path (2D linear TFI seeding + Winslow/Poisson elliptic block
 * smoothing with inter-block coupling).
 *
 * Every entry point replaces one Zig-level seam of the reference (pascalPost/turbomesh);
 * the seam is cited next to the declaration as reference file:line.  The Zig binding a
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; all host arrays are caller-owned, never retained past
 *     the call (handle API excepted: it keeps DEVICE copies only);
 *   - coordinates are Vec2d = struct{data:[2]f64} slices reinterpret-cast to double*
 *     (reference src/core/types.zig:16-37): x,y interleaved, node (i,j) at i*nj + j
 *     (types.zig:94-96), 16 B per node;
 *   - every function returns int: 0 = ok, >0 = warning status, <0 = tm_error; the message is
 *     available from tm_last_error() (pattern: cg_get_error, reference src/core/cgns.zig:19-22);
 *     nothing aborts the process;
 *   - not re-entrant per handle; no hidden global state besides the HIP runtime;
 *   - measurement and diagnostic entry points (event timing of the dominant kernel, the STREAM-style probe, the White math probe,
 *     how a handle orders its two queues) are declared in tm_hip_diag.h, not here: a binding needs none of them.
 */
#ifndef TM_HIP_H
#define TM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_HIP_ABI_VERSION 1

/* ------------------------------------------------------------------ errors */
typedef enum tm_error {
    TM_OK = 0,
    TM_W_NOT_CONVERGED = 1,   /* warning only, like BiCGStab.zig:368-369 / GMRES.zig:422 */
    TM_E_SIZE = -1,           /* error.InconsistentSize (tfi.zig:30)                       */
    TM_E_TOPOLOGY = -2,       /* asserts of smooth.zig:562, 627-631, 719; boundary.zig:280 */
    TM_E_MISMATCH = -3,       /* connectionDataCheck panic (smooth.zig:220-275)            */
    TM_E_OVERFLOW = -4,       /* BoundedArray overflow (smooth.zig:1191-1206)              */
    TM_E_UNSUPPORTED = -5,    /* error.ExternalSolverNotEnabled (solver.zig:48,56,82,89)   */
    TM_E_ARG = -6,
    TM_E_MEMORY = -7,
    TM_E_HIP = -8,            /* HIP runtime failure / no gfx950 device                    */
    TM_E_COMM = -9            /* exchange / all-reduce hook failed                         */
} tm_error;

const char* tm_last_error(void);
int tm_abi_version(void);

/* The reference logs two lines per outer iteration (std.log scope .smoothing): "iteration: {n}" before the fill and
 * "\tresidual: {(sum dx^2 + sum dy^2)^2}" after the solve (reference src/core/smoothing/smooth.zig:105, 136-137).  A caller
 * that wants them installs a sink: what = 0 announces iteration `iteration`, what = 1 carries its residual in `value`.
 * Process-wide like the reference's logger; NULL (the default) switches it off.  With a sink installed every outer
 * iteration ends with a reduction and a host round trip -- in TM_INNER_RELAX mode (one sweep = one outer iteration)
 * that means one sweep per kernel pass and one synchronisation per sweep, so install it for diagnosis, not for speed. */
typedef void (*tm_log_fn)(void* ctx, int32_t what, uint64_t iteration, double value);
void tm_set_log(tm_log_fn sink, void* ctx);

/* ------------------------------------------------------------------ data model
 * POD mirrors of the reference types, enum values in declaration order. */
enum { TM_SIDE_I_MIN = 0, TM_SIDE_I_MAX = 1, TM_SIDE_J_MIN = 2, TM_SIDE_J_MAX = 3 };   /* boundary.zig:8-13   */
enum { TM_BC_WALL = 0, TM_BC_INLET = 1, TM_BC_OUTLET = 2 };                              /* boundary.zig:178-182 */

typedef struct tm_range {        /* boundary.zig:15-19 */
    uint64_t block;
    uint32_t side;
    uint32_t _pad;
    uint64_t start, end;         /* inclusive; start > end = reversed */
} tm_range;

typedef struct tm_connection {   /* boundary.zig:119-123 */
    tm_range r[2];
    int32_t has_periodicity;     /* periodicity: ?Vec2d */
    int32_t _pad;
    double periodicity[2];       /* maps range[0] onto range[1] */
} tm_connection;

typedef struct tm_condition {    /* boundary.zig:184-187 */
    tm_range range;
    uint32_t kind;
    uint32_t _pad;
} tm_condition;

typedef struct tm_block {        /* discrete.zig:138-141 + types.zig:78-80 */
    double* xy;                  /* ni*nj*2 doubles, mutated in place by the smoother */
    uint64_t ni, nj;             /* size[0], size[1] */
} tm_block;

typedef struct tm_mesh_desc {    /* discrete.zig:166-171 (names omitted) */
    tm_block* blocks;
    uint64_t nblocks;
    tm_connection* conns;
    uint64_t nconns;
    tm_condition* bcs;
    uint64_t nbcs;
} tm_mesh_desc;

/* wall_control_function.zig:10-20, 56-68 */
enum { TM_CF_LAPLACE = 0, TM_CF_WHITE = 1 };
typedef struct tm_control_fn {
    int32_t kind;
    int32_t _pad;
    double ds_target;
    double theta_target;
} tm_control_fn;

/* solver.zig:10-27: Tag in declaration order + the new `hip` member.  Only TM_SOLVER_HIP is
 * served by this library; the reference's own tags return TM_E_UNSUPPORTED exactly like a
 * build without -Duse-umfpack returns error.ExternalSolverNotEnabled. */
enum { TM_SOLVER_GMRES = 0, TM_SOLVER_BICGSTAB = 1, TM_SOLVER_UMFPACK = 2, TM_SOLVER_PETSC = 3, TM_SOLVER_HIP = 4 };
/* inner strategy of the hip solver */
enum {
    TM_INNER_BICGSTAB = 0,   /* Picard outer iteration (smooth.zig:104-154), each frozen-coefficient system solved by
                                matrix-free BiCGStab on the row-equilibrated operator D^-1 A (replaces BiCGStab.zig:279-370) */
    TM_INNER_RELAX = 1,      /* every outer iteration is ONE fused Jacobi elliptic sweep X <- X + omega D^-1 (b - A(X) X)   */
    TM_INNER_MG_BICGSTAB = 2,/* TM_INNER_BICGSTAB, right-preconditioned by one geometric-multigrid V(2,2) cycle per block
                                (damped Jacobi, full weighting, rediscretised Winslow operator on vertex-coarsened levels;
                                the perimeter rows are applied to the interior corrections behind the cycles).  Same Picard iterates, far fewer inner iterations on
                                large blocks: what makes "to 1e-8 residual at 4096^2" practical               */
    TM_INNER_GMRES = 4,      /* the Picard outer iteration with the reference's OTHER Krylov solver restated on the device: restarted GMRES(30),
                                left-preconditioned with the diagonal, modified Gram-Schmidt Arnoldi, Givens rotations (GMRES.zig:300-423,
                                510-524; restart 30 as GMRES.zig:21).  Same scale-aware stop test as TM_INNER_BICGSTAB.  What a front end
                                maps "solver": {"gmres": {"preconditioner": "diagonal"}} of an input file to; ILU(0) (GMRES.zig:199-298) is a
                                sequential recurrence with no device counterpart                                                      */
    TM_INNER_AUTO = 3        /* resolved when the handle is created, from the mesh alone (every rank of a job decides alike):
                                TM_INNER_MG_BICGSTAB when the largest block has >= 100 000 nodes, TM_INNER_BICGSTAB below -- on the
                                reference's example meshes (T106 / LS89: 8 blocks of 10^2..10^4 nodes) every multigrid level is a handful
                                of launch-bound kernels and the plain solver is 7x faster in wall time; from ~300^2 nodes per block on
                                the cycle's O(1) iteration count wins; blocks that no connection couples take
                                the cycle from 1000 nodes on (it is block-local: a lone block needs ~25 iterations at any size, coupled
                                blocks hundreds).  A single-process handle (no rank hooks)
                                also looks at the coordinates it is given: where the cells' aspect ratio varies strongly INSIDE a block
                                (standard deviation of log(|x_xi|^2 / |x_eta|^2) over the block's nodes above 1: boundary-layer
                                clustering) the point-Jacobi cycle is a poor preconditioner and the plain solve is chosen at any size.
                                Same Picard iterates either way.                                                                    */
};
/* tm_solver_opt.flags */
enum {
    TM_OPT_SINGLE_SWEEP = 1, /* TM_INNER_RELAX: one kernel pass per sweep.  Default (bit clear): sweeps are taken three per pass
                                on blocks whose perimeter rows are all `fixed`, two per pass on coupled blocks (same arithmetic,
                                bit-identical coordinates, a third / half of the HBM traffic) */
    TM_OPT_EAGER_SCALARS = 2, /* Krylov modes: the textbook launch sequence of BiCGStab.zig:279-370 -- one kernel per vector update, one
                                scalar-update launch per reduction.  Default (bit clear) on single-process handles: the vector updates
                                are formed inside the two operator applications (two kernels per iteration; rho from r_hat.s -
                                omega r_hat.t, equal in exact arithmetic) and, on small meshes, the scalar steps travel with the
                                kernels that consume them.  Same method, iterates equal to rounding */
    TM_OPT_PRECOND_ILU0 = 8, /* tm_csr_solve ONLY (the slot that has the assembled matrix): ILU(0) -- the reference's second preconditioner
                                (preconditioner.zig:1-4; BiCGStab.zig:178-277, 384-422) -- as right preconditioner of the device BiCGStab: factor
                                and substitutions level by level on the device, bit-identical to the reference's recurrence.  The matrix-free
                                entry points answer TM_E_UNSUPPORTED: they never assemble a matrix to factorise                        */
    TM_OPT_RTOL_INITIAL = 4  /* Krylov modes: `rtol` is relative to the INITIAL residual of each inner solve (inexact Picard: stop at
                                ||D^-1(b-Ax)|| <= max(atol, rtol ||D^-1(b-A x0)||), rtol = 0 -> 1e-2) instead of ||D^-1 b||.  Every solve then
                                does work in proportion to what is left -- same fixed point, several times fewer inner iterations on the way
                                (tools/converge_probe.py); the Picard ITERATES are no longer the exact-solve ones, so parity per iterate is
                                a statement about the default mode only */
};
typedef struct tm_solver_opt {
    int32_t tag;             /* TM_SOLVER_* */
    int32_t inner;           /* TM_INNER_* */
    double rtol;             /* stop the inner solve at ||D^-1(b-Ax)||_2 <= max(atol, rtol*||D^-1 b||_2); 0 -> size-aware default: 7.5e-9 / nodes clamped to [1e-16, 1e-14] (1e-14 up to 866^2 nodes; the error of the solve is conditioning x residual and the conditioning grows with the mesh); 1e-14 at every size with the multigrid preconditioner, which makes the conditioning O(1) */
    double atol;             /* 0 -> 0 */
    uint64_t max_inner;      /* BiCGStab iteration cap per Picard solve; 0 -> max(10000, 12 sqrt(nodes)) (the reference caps at 1000, BiCGStab.zig:19, with its far looser stop test) */
    uint32_t check_every;    /* host convergence poll interval in inner iterations; 0 -> 8 (1 with the multigrid preconditioner) */
    uint32_t flags;          /* TM_OPT_* bits; 0 = defaults */
    double omega;            /* relaxation factor of TM_INNER_RELAX; 0 -> 1.0 */
} tm_solver_opt;

typedef struct tm_stats {
    uint64_t outer_iterations;
    uint64_t inner_iterations;   /* BiCGStab iterations summed over the outer iterations (both components advance together) */
    uint64_t operator_sweeps;    /* applications of the 9-point operator to the whole mesh */
    double last_residual;        /* (sum dx^2 + sum dy^2)^2 of the last outer iteration, smooth.zig:136 */
    double last_dx2, last_dy2;   /* sum (x_old-x_new)^2, sum (y_old-y_new)^2, smooth.zig:112-134 */
    double scaled_residual_rms;  /* sqrt(||D^-1(b - A(X)X)||^2 / (2*dof)) at the start of the last outer iteration */
    double seconds;              /* wall time of the call, like smooth.zig:156-160 */
    int32_t not_converged;       /* inner solves that hit max_inner */
    int32_t _pad;
} tm_stats;

/* ------------------------------------------------------------------ seam 3: TFI
 * Replaces tfi.linear2dBoundaryBlendedControlFunction (reference src/core/tfi.zig:112-208),
 * called from Block2d.init (src/core/discrete.zig:146-157).  Host pointers; xy_out[ni*nj*2]
 * is overwritten for ALL nodes (boundary nodes included, tfi.zig:164-205).
 * s1,s2 [ni] / t1,t2 [nj]: clusterings of the i_min,i_max / j_min,j_max edges, first 0, last
 * exactly 1 (tfi.zig:135-145) else TM_E_ARG; corner points must agree within 1e-10
 * (tfi.zig:150-162) else TM_E_MISMATCH. */
int tm_tfi_block(double* xy_out, uint64_t ni, uint64_t nj,
                 const double* x_i_min, const double* x_i_max,   /* ni*2 each */
                 const double* x_j_min, const double* x_j_max,   /* nj*2 each */
                 const double* s1, const double* s2, const double* t1, const double* t2);

/* Replaces tfi.linear2d (reference src/core/tfi.zig:19-67): plain TFI, xi=i/(ni-1), eta=j/(nj-1),
 * corners from the i edges. */
int tm_tfi_linear2d(double* xy_out, uint64_t ni, uint64_t nj,
                    const double* e_i_min, const double* e_i_max, const double* e_j_min, const double* e_j_max);

/* ------------------------------------------------------------------ seam 1: whole smoother
 * Replaces smooth.mesh(allocator, *Mesh, iterations, solver.Option, Algorithm)
 * (reference src/core/smoothing/smooth.zig:74-80); callers gui/main.zig:52, wasm/lib.zig:46-52.
 * Mutates mesh->blocks[b].xy in place.  iterations == 0 is legal and leaves the mesh untouched. */
int tm_smooth_mesh(const tm_mesh_desc* mesh, uint64_t iterations, const tm_solver_opt* opt,
                   const tm_control_fn* cf, tm_stats* stats /* may be NULL */);

/* ------------------------------------------------------------------ seam 2: the linear-solver slot
 * Replaces a backend of solver.Solver (reference src/core/smoothing/solver.zig:40-93; pattern umfpack.zig:18-55): the
 * caller has ASSEMBLED RowCompressedMatrixSystem2d (smooth.zig:277-307) and hands over its CSR arrays -- Ap = lhs_p[n+1],
 * Ai = lhs_i, values = lhs_values -- with both right-hand sides and the solution vectors x_new / y_new, which carry the
 * initial guess in and the solution out (warm start, BiCGStab.zig:136-153).  The x- and the y-system share the pattern and
 * differ in the two entries of every sliding row (system.fillXSpecific / fillYSpecific, smooth.zig:1115-1165): pass the
 * values after fillXSpecific as Ax_x and a copy taken after fillYSpecific as Ax_y (NULL = same values for both; exact
 * whenever the mesh has no inlet / outlet condition).  Both components are solved together on the device by BiCGStab
 * -- or, with opt->inner == TM_INNER_GMRES, by the reference's restarted GMRES(30) (GMRES.zig:300-423), left-preconditioned; with
 * TM_OPT_PRECOND_ILU0 in opt->flags either of them takes ILU(0) instead of the diagonal: the four combinations of solver.zig:18-27 --
 * on D^-1 A with the stop test of tm_solver_opt (rtol, atol, max_inner, check_every; opt may be NULL = defaults; rtol 0 -> 1e-14 here at
 * every size: the size-aware default belongs to the matrix-free path's own operator).  A recurrence residual that fails to halve over
 * max(4000, 4 sqrt(n)) iterations ends the solve as not converged.
 * Returns TM_OK, TM_W_NOT_CONVERGED (warning, like BiCGStab.zig:368-369) or a negative error; host pointers throughout;
 * the matrix is uploaded per call -- this is the faithful "reference-assembled, GPU-solved" mode, not the fast path
 * (that is seam 1, which never assembles a matrix). */
int tm_csr_solve(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax_x, const double* Ax_y /* may be NULL */,
                 const double* bx, const double* by, double* x /* in: guess, out */, double* y, const tm_solver_opt* opt /* may be NULL */,
                 tm_stats* stats /* may be NULL */);

/* ------------------------------------------------------------------ persistent handle
 * Same smoother with the coordinates resident in HBM between calls (for callers that iterate,
 * inspect, iterate ...).  create uploads the mesh; iterate runs `iterations` outer iterations on
 * the device; download writes the current coordinates back into the caller's arrays. */
typedef struct tm_smoother tm_smoother;

/* Optional hooks that make ONE rank of a multi-GPU job out of a handle (one process per GPU):
 * the mesh description handed to create() is the GLOBAL topology, `owner[b]` names the rank
 * that owns block b, only owned blocks need coordinates.  exchange() must fill recv_buf (device)
 * from the peers' send_buf (device) according to the plan returned by tm_smoother_exchange_plan;
 * allreduce_sum() sums n doubles in place (device) over all ranks.  Every call names the stream it must be ordered on:
 * the handle's stream, or -- for the exchanges of relaxation sweep pairs -- a second stream the handle owns, on which the
 * exchanges and perimeter rows run beside the interior pass. */
typedef struct tm_comm_hooks {
    void* ctx;
    int32_t rank, nranks;
    const int32_t* owner;   /* [nblocks] */
    int (*exchange)(void* ctx, const double* send_buf, double* recv_buf, void* stream);
    int (*allreduce_sum)(void* ctx, double* buf, int32_t n, void* stream);
    /* Optional split form: when exchange_wait != NULL, exchange() only STARTS the transfer (it may return before the
     * ghost rows have arrived) and exchange_wait() makes `stream` wait for it.  The library then runs the interior-row
     * kernel K2 -- which never reads ghost rows -- between the two, hiding the transfer behind it. */
    int (*exchange_wait)(void* ctx, void* stream);
    /* Optional caller-provided device memory (e.g. a torch tensor, so the hooks can hand views of it to
     * torch.distributed): when workspace != NULL every device buffer of the handle is carved from it.
     * Size it with tm_smoother_workspace_bytes. */
    void* workspace;
    uint64_t workspace_bytes;
} tm_comm_hooks;
/* Device bytes a handle for this mesh/options/partition needs (hooks may be NULL = single process). */
int tm_smoother_workspace_bytes(const tm_mesh_desc* mesh, const tm_solver_opt* opt, const tm_control_fn* cf,
                                const tm_comm_hooks* hooks, uint64_t* bytes);

int tm_smoother_create(const tm_mesh_desc* mesh, const tm_solver_opt* opt, const tm_control_fn* cf,
                       const tm_comm_hooks* hooks /* NULL = single process */, void* stream /* hipStream_t or NULL */,
                       tm_smoother** out);
int tm_smoother_iterate(tm_smoother* s, uint64_t iterations, tm_stats* stats);
/* Outer iterations until the scaled nonlinear residual stats->scaled_residual_rms = sqrt(||D^-1 (b - A(X) X)||^2 / (2 dof)) is
 * <= scaled_residual_tol, at most max_iterations of them (the reference iterates a fixed count from its input file,
 * smooth.zig:104; it has no stop test).  TM_OK = reached; TM_W_NOT_CONVERGED = max_iterations hit or an inner solve did not
 * converge.  stats->outer_iterations = iterations actually performed (Picard: solves; relax: sweeps, tested every 32). */
int tm_smoother_iterate_until(tm_smoother* s, uint64_t max_iterations, double scaled_residual_tol, tm_stats* stats);
/* ... until the update of the last outer iteration, sqrt((sum dx^2 + sum dy^2) / nodes) over the whole mesh, is <= update_rms_tol:
 * the quantity the reference forms and logs per iteration (smooth.zig:112-137) -- convergence of the COORDINATES.  Returns
 * TM_W_NOT_CONVERGED when max_iterations ran out first (stats are valid either way). */
int tm_smoother_iterate_until_update(tm_smoother* s, uint64_t max_iterations, double update_rms_tol, tm_stats* stats);
int tm_smoother_download(tm_smoother* s, const tm_mesh_desc* mesh);
int tm_smoother_upload(tm_smoother* s, const tm_mesh_desc* mesh);
void tm_smoother_destroy(tm_smoother* s);

/* Built-in transport for the hooks above: RCCL point-to-point (xGMI) + all-reduce issued by the library itself, so that no
 * host-language callback sits in the sweep loop.  librccl is dlopen'ed: pass the path of the copy the process already uses
 * (e.g. <torch>/lib/librccl.so) or NULL for the default search.  Rank 0 calls tm_rccl_unique_id and hands the 128 bytes
 * to the other ranks by whatever rendezvous the host has (torch.distributed broadcast, MPI, a file); every rank then calls
 * tm_rccl_comm_create (collective) with the current HIP device set.  tm_rccl_hooks fills `hooks` for a mesh partition:
 * exchange/exchange_wait = grouped ncclRecv/ncclSend per neighbouring rank on a side stream, fenced with events against
 * the handle's stream; allreduce_sum = ncclAllReduce.  The hooks (and hooks->owner) stay valid while `comm` lives. */
typedef struct tm_rccl_comm tm_rccl_comm;
#define TM_RCCL_ID_BYTES 128
int tm_rccl_unique_id(const char* librccl_path, void* id_out /* TM_RCCL_ID_BYTES */);
int tm_rccl_comm_create(const char* librccl_path, const void* id, int32_t rank, int32_t nranks, tm_rccl_comm** out);
void tm_rccl_comm_destroy(tm_rccl_comm* comm);
int tm_rccl_hooks(tm_rccl_comm* comm, const tm_mesh_desc* mesh, const int32_t* owner /* [nblocks] */, tm_comm_hooks* hooks);
/* The same when the options of the handle the hooks are for are known (pass the very structures given to tm_smoother_create): a handle that
 * never runs sweep triples -- the Krylov modes, the White control function, TM_OPT_SINGLE_SWEEP -- then exchanges the depth-2 halo only
 * (tm_rccl_hooks sizes its tables by the topology alone: depth 3 wherever every block has at least 16 x 16 nodes, whatever the handle does with it).
 * A handle created with the library's own hooks follows the depth their tables were built for. */
int tm_rccl_hooks_for(tm_rccl_comm* comm, const tm_mesh_desc* mesh, const int32_t* owner, const tm_solver_opt* opt, const tm_control_fn* cf /* NULL = laplace */,
                      tm_comm_hooks* hooks);

/* The table tm_rccl_hooks builds for a rank, host-only (no communicator, no GPU): per neighbouring rank the offset and count
 * (rows of 16 B) handed to ncclSend / ncclRecv in one group per exchange.  Pairwise symmetry -- send_cnt of a towards b ==
 * recv_cnt of b from a, peers mutual, offsets inside send_rows / recv_rows -- is what keeps ncclGroupEnd from hanging; a caller
 * (or a test) can check it for every rank of a partition before any process joins a communicator.
 * send_off indexes the rank's own vector when direct_send (n_owned + n_ghost rows), the packed send buffer otherwise. */
typedef struct tm_rccl_peer_table {
    int32_t npeers, direct_send;
    int64_t send_rows, recv_rows;   /* extent of the buffers the offsets index */
    int32_t* peer;                  /* [npeers] ascending                      */
    int64_t *send_off, *send_cnt, *recv_off, *recv_cnt;   /* [npeers] each      */
} tm_rccl_peer_table;
int tm_rccl_peer_table_build(const tm_mesh_desc* mesh, const int32_t* owner, int32_t rank, int32_t nranks, tm_rccl_peer_table* out);
void tm_rccl_peer_table_free(tm_rccl_peer_table* table);

/* Exchange plan of a handle created with hooks: npeers peers; for peer k, send_count[k] rows
 * (16 B each) start at send_offset[k] rows into send_buf, same for recv.  Rows are double2.
 * send_buf is a packed buffer -- or, when every peer's rows are one contiguous run of the rank's vector (interfaces along
 * whole block rows), the vector itself, with send_offset[k] the run's first row: no pack kernel runs then.  Either way the
 * hook sends send_count[k] rows from send_buf + send_offset[k]. */
int tm_smoother_exchange_plan(const tm_smoother* s, int32_t* npeers, const int32_t** peer_rank,
                              const int64_t** send_offset, const int64_t** send_count,
                              const int64_t** recv_offset, const int64_t** recv_count);

/* Introspection used by the parity tests (and by a Zig caller that wants the operator only):
 * out = A(X) * in over the rank's owned rows, A assembled matrix-free from the CURRENT device
 * coordinates; scaled != 0 applies the row equilibration D^-1.  in/out are host arrays with
 * 2*dof doubles in global row order (smooth.zig:1643-1645).  Single-process handles only. */
int tm_smoother_apply(tm_smoother* s, const double* in_xy, double* out_xy, int scaled);
/* The system of the CURRENT device coordinates as the reference ASSEMBLES it -- RowCompressedMatrixSystem2d (smooth.zig:277-385: lhs_p,
 * lhs_i in the reference's column order) filled by system.fill (smooth.zig:923-1113) on the device: interior rows StencilData.init's nine
 * values in the reference's expression order (smooth.zig:171-216), perimeter rows from the plan; Ax_x / Ax_y = lhs_values after fillXSpecific
 * / fillYSpecific (smooth.zig:1115-1165).  Ap [dof + 1], Ai / Ax_x / Ax_y [nnz]; any of them may be NULL (all NULL: *nnz only); the
 * right-hand sides come from tm_smoother_rhs.  What a Zig caller gets from its own system.fill -- the arrays tm_csr_solve takes -- and what
 * the parity tests compare with the faithful oracle bit for bit.  Single-process handles only. */
int tm_smoother_assemble_csr(tm_smoother* s, int32_t* Ap, int32_t* Ai, double* Ax_x, double* Ax_y, uint64_t nnz_capacity, uint64_t* nnz);
/* out = A(X) * in evaluated THROUGH that assembled system: every row the sum of its products in CSR order, un-fused -- the reference's own
 * mat-vec (BiCGStab.zig:424-435) bit for bit, interior rows included.  (tm_smoother_apply is the matrix-free fast path: the same perimeter
 * rows, interior rows in a factored form within 16 eps sum |c_k w_k| of this.)  Single-process handles only. */
int tm_smoother_apply_reference_order(tm_smoother* s, const double* in_xy, double* out_xy);
/* Right-hand side b (2*dof doubles) for the current coordinates (smooth.zig:780-921, 1060-1061). */
int tm_smoother_rhs(tm_smoother* s, double* rhs_xy);
/* Row kind per global row: -1 interior, else BlockBoundaryPointKind (smooth.zig:1168-1174). */
int tm_smoother_row_kinds(const tm_smoother* s, int32_t* kinds /* [dof] */);
uint64_t tm_smoother_dof(const tm_smoother* s);
/* The inner strategy the handle runs (TM_INNER_*): what TM_INNER_AUTO resolved to, else the option as given. */
int tm_smoother_inner(const tm_smoother* s);
/* Current control function (P,Q) per node, 2*dof doubles (wall_control_function.zig:22-54). */
int tm_smoother_control_function(tm_smoother* s, double* pq);

/* ------------------------------------------------------------------ export (the step right after the path)
 * The reference's structured output hands the writer one plane per coordinate with i fastest: plane[j*ni + i] =
 * block(i,j) (reference src/core/cgns.zig:75-104, called from discrete.zig:197-216 Mesh.write and smooth.zig:396-414);
 * optionally the control function as planes "P" and "Q" (cgns.zig:106-154).  Both entry points do that de-interleaving
 * transpose on the device.  tm_export_soa: host block in, host planes out.  tm_smoother_export_soa: planes of an owned
 * block from the coordinates resident in the handle; p, q may both be NULL (laplace exports zeros). */
int tm_export_soa(const double* xy /* ni*nj*2 */, uint64_t ni, uint64_t nj, double* x_out /* ni*nj */, double* y_out /* ni*nj */);
int tm_smoother_export_soa(tm_smoother* s, uint64_t block, double* x, double* y, double* p, double* q);

/* ------------------------------------------------------------------ host-only planning (no GPU needed)
 * The perimeter-row table the device kernels consume, exported as CSR so it can be compared with
 * the reference's assembled rows (smooth.zig:421-921).  Only sizes/topology are read from `mesh`
 * (coordinates may be NULL).  Arrays are malloc'ed by the library; free with tm_plan_free. */
typedef struct tm_plan_rows {
    uint64_t nrows;          /* number of perimeter rows of the whole mesh                         */
    int64_t* row;            /* [nrows] global row id, ascending                                   */
    int32_t* kind;           /* [nrows] BlockBoundaryPointKind                                     */
    int32_t* ncols;          /* [nrows]                                                            */
    int64_t* cols;           /* [nrows*9] global column ids, ascending                             */
    double* coef_x;          /* [nrows*9] x-system coefficients (NaN for `smoothed` rows: dynamic) */
    double* coef_y;          /* [nrows*9] y-system coefficients                                    */
    double* rhs;             /* [nrows*2] static rhs (NaN where it is taken from the coordinates)  */
    int32_t* slot;           /* [nrows*9] smoothed rows: StencilData index (smooth.zig:175-185) per column */
} tm_plan_rows;
int tm_plan_build(const tm_mesh_desc* mesh, tm_plan_rows* out);
void tm_plan_free(tm_plan_rows* rows);

/* Rank-local view of a partitioned mesh (host-only): which rows a rank owns, which remote rows it keeps copies of
 * (ghost rows, appended after the owned rows in every rank-local vector) and the halo exchange lists.
 * Every rank computes the same tables from the global topology, so no set-up communication is needed.
 * The halo is two rows deep: the remote rows the rank's perimeter rows read (the reference couples blocks through
 * smooth.zig:618-693, 994-1105: interface row, first interior row of either side, junction neighbours) -- listed in
 * ghost_row_* with their own columns -- AND the remote rows those read, so that a rank can evaluate the former one sweep
 * ahead and a PAIR of relaxation sweeps needs one exchange. */
typedef struct tm_plan_local_info {
    int64_t n_owned, n_ghost, n_send;
    int32_t npeers, nowned_blocks;
    int64_t* owned_blocks;   /* [nowned_blocks] global block ids, ascending                      */
    int64_t* local_start;    /* [nowned_blocks] rank-local index of node (0,0) of each of them   */
    int64_t* ghost_gid;      /* [n_ghost] global row ids, grouped by owner rank then ascending   */
    int32_t* send_ids;       /* [n_send] rank-local indices of the rows packed for the peers     */
    int64_t* send_gid;       /* [n_send] their global row ids                                    */
    int32_t* peer_rank;      /* [npeers]                                                         */
    int64_t* send_offset;    /* [npeers] rows; peer k gets send_ids[send_offset[k] .. +send_count[k]) */
    int64_t* send_count;
    int64_t* recv_offset;    /* [npeers] rows into the ghost segment                             */
    int64_t* recv_count;
    int64_t* send_first;     /* [npeers] first rank-local row of peer k's send list                            */
    int32_t direct_send;     /* 1: every peer's send list is one ascending run of local rows -- a handle then sends
                                straight from the vector (tm_smoother_exchange_plan offsets = send_first), no pack kernel */
    int32_t _pad;
    int64_t n_ghost_rows;    /* ghost rows this rank can evaluate itself (the depth-1 part of the halo)                 */
    int64_t* ghost_row_gid;  /* [n_ghost_rows]                                                                           */
    int32_t* ghost_row_kind; /* [n_ghost_rows] BlockBoundaryPointKind of the row, or 5 = interior node of the remote block */
    int64_t* ghost_row_cols; /* [n_ghost_rows * 9] global ids of its columns, -1 padded                                  */
} tm_plan_local_info;
int tm_plan_local(const tm_mesh_desc* mesh, const int32_t* owner, int32_t rank, int32_t nranks, tm_plan_local_info* out);
void tm_plan_local_free(tm_plan_local_info* info);

/* ------------------------------------------------------------------ device-level entry points
 * Same kernels on caller-provided DEVICE pointers and stream, for callers that keep blocks in
 * HBM (bench.py, the multi-GPU driver).  No allocation, no synchronisation. */
int tm_dev_tfi_block(double* d_xy, uint64_t ni, uint64_t nj,
                     const double* d_x_i_min, const double* d_x_i_max, const double* d_x_j_min, const double* d_x_j_max,
                     const double* d_s1, const double* d_s2, const double* d_t1, const double* d_t2, void* stream);
/* One fused Jacobi elliptic sweep of a single block with fixed boundary (Laplace control function):
 * d_out = d_in + omega * D^-1 (b - A(d_in) d_in) on interior nodes, boundary nodes copied;
 * d_partials[2*nwg] receives per-workgroup sums of (dx^2, dy^2); returns nwg through *nwg. */
int tm_dev_relax_sweep(const double* d_in, double* d_out, uint64_t ni, uint64_t nj, double omega,
                       double* d_partials, uint64_t partials_capacity, uint64_t* nwg, void* stream);
uint64_t tm_dev_relax_partials_needed(uint64_t ni, uint64_t nj);

#ifdef __cplusplus
}
#endif
#endif /* TM_HIP_H */
