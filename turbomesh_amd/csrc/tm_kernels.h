// Launch interface of the gfx950 kernels (internal to libtm_hip.so).
// Kernel ids follow SURVEY.md section 7: K1 tfi_blend, K2 winslow_apply, K3 fused Krylov vector
// kernels, K4/K5 perimeter rows, K6 white control function, K7 residual + copy-back.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace tmh {

// ---- K2 modes: what is written for an interior row with sum = (A(xk) in)_row, dinv = 1/a_ii
enum ApplyMode : int {
    MODE_RAW = 0,      // out = sum
    MODE_SCALED = 1,   // out = sum * dinv                      (D^-1 A in)
    MODE_RESID = 2,    // out = rhs*dinv - sum*dinv             (D^-1 (b - A in)), rhs = 0 on interior rows
    MODE_RELAX = 3,    // out = in + omega * (rhs*dinv - sum*dinv)
    // diagnostics (tools/tune_k2.py only): same tiling and data movement, reduced arithmetic
    MODE_DIAG_COPY = 4,   // out = in
    MODE_DIAG_SUM9 = 5,   // out = plain sum of the 9 neighbours (loads, halo, lane shifts; no coefficients)
    MODE_DIAG_NOSTORE = 6,   // full relax arithmetic, all loads, NO store (result folded into the partial sums)
    MODE_DIAG_NOLOAD = 7,    // full relax arithmetic and stores, rows are NOT re-loaded (window reused)
    MODE_DIAG_MATH = 8,      // full relax arithmetic only: no loads, no stores
    // multigrid (block-local error equation D^-1 A e = f, e = 0 on the perimeter; f arrives through `aux`)
    MODE_MG_RESID = 9,     // out = f - D^-1 A in
    MODE_MG_SMOOTH = 10,   // out = in + omega * (f - D^-1 A in)        (damped Jacobi)
    MODE_MG_FIRST2 = 11    // out = omega * (2 in - omega * D^-1 A in): the first TWO damped-Jacobi sweeps from a zero guess, in = f
};
// ---- fused partial reductions written per workgroup (x and y components separately)
enum DotMode : int {
    DOT_NONE = 0,
    DOT_AUX = 1,    // [aux.out]_x, [aux.out]_y                       (sigma = r_hat . v)
    DOT_IN = 2,     // [in.out]_x, [in.out]_y, [out.out]_x, [out.out]_y   (t.s, t.t)
    DOT_OUT2 = 3,   // [out.out]_x, [out.out]_y                       (||r||^2)
    DOT_DELTA = 4,  // [(out-in)^2]_x, [(out-in)^2]_y                 (relax: sum dx^2, dy^2)
    DOT_AUX2 = 5,   // [aux.out]_x, [aux.out]_y, [out.out]_x, [out.out]_y (t.s, t.t when the operator acted on a preconditioned vector)
    DOT_IN_SS = 6,  // DOT_IN + [in.in]_x, [in.in]_y   (t.s, t.t, ||s||^2: the second apply of an iteration acting on s = r - alpha v formed on the fly)
    DOT_B2 = 7      // DOT_IN + [aux.in]_x,y + [aux.out]_x,y   (t.s, t.t, r_hat.s, r_hat.t: the two-kernel iteration, STEP_B2)
};
constexpr int MAX_PARTIALS = 8;
struct KrylovScalars;
struct LazyScalars;

struct ApplyBlock {
    // virtual-input kernels (launch_apply_virtual): the operator acts on a vector formed on the fly from in, in2, in3
    const double2* in2 = nullptr;
    const double2* in3 = nullptr;
    double2* pout = nullptr;        // VK_P, VK_R: where the formed vector is stored (owned rows)
    const double2* in4 = nullptr;   // VK_R: p
    double2* rout = nullptr;        // VK_R: the updated residual (owned rows)
    double2* uio = nullptr;         // VK_R: the solution vector, updated in place (owned rows)
    int rows = 0;                   // rows per chunk; 0 = the per-block rule (rows_per_chunk, tm_kernels.hip)
    // VK_PRO (launch_mg_prolong_smooth): in2 = the coarse correction, (nic, njc) its size, ci / cj = which directions were coarsened
    int mg_nic = 0, mg_njc = 0, mg_ci = 0, mg_cj = 0;
    const double2* in;    // vector the operator acts on, pointing at the block's node (0,0)
    const double2* xk;    // frozen coordinates the coefficients are built from (== in for field mode)
    const double2* pq;    // control function (P,Q) or nullptr (Laplace)
    const double2* aux;   // second vector for DOT_AUX or nullptr
    double2* out;
    int ni, nj;
    double omega;
    double* partials;     // [nwg * MAX_PARTIALS] for this launch
};
// number of workgroups k2 launches for a block (also the number of partial rows it writes)
int apply_block_nwg(int ni, int nj, int rows = 0);
int apply_block_nwg_overlap(int ni, int nj, int rows = 0);   // the same for the overlapping-strip layout of the large-mesh Krylov kernels
void tune_apply(int rows_per_chunk, int unroll, int pipe, int nt);   // <=0 (pipe, nt: <0) keeps the current value
hipError_t launch_apply_block(const ApplyBlock& a, int mode, int dot, hipStream_t stream);
// several blocks of one rank in one launch (per group of APPLY_BATCH_MAX); block k's partial sums go to blocks[k].partials
constexpr int APPLY_BATCH_MAX = 8;
struct ApplyBatch {
    ApplyBlock b[APPLY_BATCH_MAX];
    int RI[APPLY_BATCH_MAX], nSG[APPLY_BATCH_MAX], nRC[APPLY_BATCH_MAX];
    int start[APPLY_BATCH_MAX];   // first workgroup of block k (unused entries: the grid size)
    int n;
};
hipError_t launch_apply_blocks(const ApplyBlock* blocks, int n, int mode, int dot, hipStream_t stream);

// ---- K2x2: TWO fused Jacobi sweeps of one block in one pass (field mode, Laplace).  `in` = X^k (all rows),
// `mid` = X^(k+1): its perimeter rows must already hold the perimeter-row kernel's result for X^k; the kernel
// reads them and WRITES the first-interior ring of X^(k+1) into it (what the next perimeter-row kernel and the
// halo exchange need); `out` = X^(k+2) interior rows.  Same arithmetic as two K2 launches, bit for bit.
struct Relax2Block {
    const double2* in;
    double2* mid;
    double2* out;
    int ni, nj;
    double omega;
    double* partials;   // [nwg * MAX_PARTIALS]: sum (X^(k+2) - X^(k+1))^2 over interior rows
    const int32_t* border;   // tile ids of the BORDER workgroups (relax2_border_tiles), device array
    int nborder;
    int store_nt = 1;   // result stores: 1 = streaming (nt), 0 = plain -- chosen by the handle from its footprint (launch_relax2_block)
    int dyn;            // sides whose perimeter values change from sweep to sweep (bit 0: i = 0, 1: i = ni-1, 2: j = 0, 3: j = nj-1);
                        // along the other sides `mid` holds constants, and workgroups touching only those count as INSIDE
};
bool relax2_supported(int ni, int nj);
int relax2_rows_per_chunk(int ni, int nj);   // rows per workgroup chosen for this block on this device (fixed at handle creation)
int relax2_block_nwg(int ni, int nj, int rows_per_chunk);
// tile ids (row chunk * strip groups + strip group) of the workgroups a BORDER launch runs for this block and `dyn` mask
std::vector<int32_t> relax2_border_tiles(int ni, int nj, int rows_per_chunk, int dyn);
// Which workgroups of the K2x2 grid a launch runs.  BORDER = every workgroup with a wave that reads perimeter values of X^(k+1)
// from `mid` or writes its first-interior ring ALONG A SIDE WHOSE PERIMETER ROWS ARE NOT ALL `fixed` (Relax2Block::dyn); INSIDE = the rest (INSIDE_A / INSIDE_B: its two halves), which read nothing but
// interior rows of X^k -- a multi-rank handle runs them while the halo exchanges are in flight.
enum Relax2Subset { R2_ALL = 0, R2_BORDER = 1, R2_INSIDE_A = 2, R2_INSIDE_B = 3, R2_INSIDE = 4 };
// lds_bytes: dynamic LDS the launch asks for.  K2x2 needs none; a multi-rank interior pass asks for a third of the CU's 160 KB to cap
// itself at 3 workgroups per CU (see Smoother::inside_lds).
// wait: the workgroups spin (one thread each, with sleeps) until *counter >= target before they touch memory -- what a
// launch_queue_wait in front of the launch would do, without the launch (border passes: a few dozen workgroups).
// Time limit of a device-side wait in ticks of the constant 100 MHz clock: a hang guard for a neighbouring rank that never shows up
// (its exchange kernel would spin for ever as well), NOT what decides whether two queues may wait for each other -- that is
// Smoother::queue_self_test, once per handle, with QUEUE_SELF_TEST_TICKS.
constexpr long long QUEUE_WAIT_TICKS = 30LL * 100000000LL;      // 30 s
constexpr long long QUEUE_SELF_TEST_TICKS = 500000LL;            // 5 ms
struct QueueWait {
    const uint32_t* counter = nullptr;
    uint32_t target = 0;
    uint32_t* error = nullptr;
    long long limit_ticks = QUEUE_WAIT_TICKS;
};
hipError_t launch_relax2_block(const Relax2Block& a, int rows_per_chunk, int dot, int subset, hipStream_t stream, size_t lds_bytes = 0,
                               const QueueWait* wait = nullptr);
struct Relax2Batch {
    Relax2Block b[8];
    int RI[8], nSG[8], nRC[8], start[8];
    int n;
};
hipError_t launch_relax2_blocks(const Relax2Block* blocks, const int* rows_per_chunk, int n, int dot, int subset, hipStream_t stream,
                                size_t lds_bytes = 0, const QueueWait* wait = nullptr);
void tune_fuse_rows(int rows);
// ---- K2x3: THREE fused sweeps per pass of blocks whose perimeter rows are all `fixed` (in -> out; `mid`, `border`, `dyn` unused)
bool relax3_supported(int ni, int nj);
int relax3_rows_per_chunk(int ni, int nj);
// ... chosen for the blocks of one launch together (n <= APPLY_BATCH_MAX; launch_relax3_blocks groups the blocks in this order)
void relax3_rows_for_launch(const int* ni, const int* nj, int n, int* rows, bool beside_chain);
int relax3_block_nwg(int ni, int nj, int rows_per_chunk);
hipError_t launch_relax3_blocks(const Relax2Block* blocks, const int* rows_per_chunk, int n, int dot, hipStream_t stream);

// ---- K4/K5 perimeter rows.  The rows of an interface are REGULAR: along a connection the row id, every column id and the
// four metric neighbours advance by constant strides while kind, column count, stencil slots and static coefficients stay
// the same (smooth.zig:518-616 fixes the column order per connection, not per point).  The host therefore compresses the
// per-row table of tm_plan into RUNS; a workgroup serves one stretch of one run, reads the run's descriptor through the
// scalar cache and computes its indices arithmetically, so the only vector loads of a row are the VALUES it gathers --
// one level of memory latency instead of two (index tables -> values), which is what a perimeter-row pass costs when it
// runs beside a bandwidth-saturating interior pass.  Irregular rows (connection end points, junctions) are runs of one.
struct EdgeRun {
    int32_t first, count;            // positions [first, first + count) of the run's rows in `rhs` (run order)
    int32_t row0, row_stride;        // local vector index of row k: row0 + k * row_stride
    int32_t col0[9], col_stride[9];  // its columns, ascending GLOBAL id order (the reference's CSR order)
    int32_t met0[4], met_stride[4];  // smoothed rows: im1_j, ip1_j, i_jm1, i_jp1
    int8_t kind, ncols, self;        // BlockBoundaryPointKind (5 = interior node of a remote block), #columns, diagonal position
    uint8_t flags;                   // bit0 periodic, bit1 swap (Q,P), bit2 / bit3: rhs_x / rhs_y = the row's current value (ghost copies), bit4: kind 5 row of an OWNED node (its displacement enters DOT_DELTA)
    int8_t slot[9];                  // smoothed rows: stencil slot per column
    int8_t _pad[3];
    double cx[9], cy[9];             // static x / y system coefficients
    double per[2];                   // periodicity of the row's connection
};
struct EdgeRowsDev {
    int nrows = 0;                      // rows in all runs
    int nwg = 0;                        // workgroups of a launch (= partial-sum rows it writes)
    const EdgeRun* runs = nullptr;
    const int32_t* wg_run = nullptr;    // [nwg] run served by workgroup w
    const int32_t* wg_k0 = nullptr;     // [nwg] first point of the run it serves
    const double* rhs = nullptr;        // [nrows*2] static right-hand side, run order
};
constexpr int EDGE_BLOCK = 128;
// The three level passes of a coupled sweep triple in ONE launch (Smoother::relax_triples_coupled): the rows of level 3 are cut into
// strips along their grid lines; a workgroup owns one strip and evaluates -- redundantly, with the very run descriptors of the three
// level tables -- every level-1 and level-2 row its strip depends on (the closure is computed on the host), level by level with a barrier
// in between.  What a workgroup reads at level l it wrote itself at level l - 1 (or nobody writes it at all), so no workgroup waits for
// another; neighbouring strips store the same bits where their closures overlap.  One kernel boundary instead of three on the
// latency-critical chain.  A task = up to 64 consecutive points of one run, taken by one wave (the descriptor stays wave-uniform).
struct LevelTask {
    int32_t run, k0, count;
};
struct FusedLevelsDev {
    int nstrips = 0;
    const LevelTask* tasks = nullptr;   // all strips, level by level
    const int32_t* off = nullptr;       // [nstrips * 4]: tasks of level l of strip s are off[4 s + l] .. off[4 s + l + 1]
};
constexpr int LEVELS_BLOCK = 512;   // (1024: no faster -- 2048^2 rank 11.9 against 10.8-11.8 us per sweep -- and twice the LDS)
hipError_t launch_edge_levels3(const FusedLevelsDev& F, const EdgeRowsDev& e1, const EdgeRowsDev& e2, const EdgeRowsDev& e3, const double2* x, double2* m,
                               double2* m2, double2* out, const double2* pq, double omega, int dot, double* partials /* [nstrips * MAX_PARTIALS] */,
                               hipStream_t stream);
// out/in/xk/pq/aux are the rank-local vectors (owned rows then ghost rows)
// signal != nullptr: the kernel's first thread bumps *signal as it starts -- "everything in front of this launch in its queue is
// complete and visible device-wide" (what launch_queue_signal would announce from a launch of its own)
hipError_t launch_edge_rows(const EdgeRowsDev& e, const double2* in, const double2* xk, const double2* pq, const double2* aux,
                            double2* out, double omega, int mode, int dot, double* partials, hipStream_t stream, uint32_t* signal = nullptr);
// interior rows of n <= APPLY_BATCH_MAX blocks + the perimeter rows in one launch (Krylov modes); hipErrorNotSupported = no such kernel
hipError_t launch_apply_edge_blocks(const ApplyBlock* blocks, int n, int mode, int dot, const EdgeRowsDev& e, const double2* in, const double2* xk,
                                    const double2* pq, const double2* aux, double2* out, double* edge_partials, hipStream_t stream);
// An apply of a BiCGStab iteration with the vector update in front of it folded in (single process):
//   kind 1 (s):  out = D^-1 A (in - alpha in2)                         s never stored; partial sums DOT_IN_SS
//   kind 2 (p):  out = D^-1 A p',  p' = in + beta (in2 - omega in3)    p' stored to pout (a different array than in2); partial sums
//                DOT_AUX with aux
// in = r; (in2, in3) = (v, -) or (p, v).  Rank-local vectors; blocks[k].in/in2/in3/pout point at the block's node (0,0).
// Interior rows of all blocks (one launch per group of APPLY_BATCH_MAX) and the perimeter rows; the scalars come from `scal`
// (pending steps applied by the first kernel that runs).
//   kind 3 (r):  the whole vector part of an iteration in front of its first apply (two-kernel iteration):
//                s = in - alpha in2, r' = s - omega in3, p' = r' + beta (in4 - omega in2), u += alpha in4 + omega s;
//                out = D^-1 A p'; r', p' stored to rout, pout (arrays other than in, in4), u in place; partial sums r_hat . out and
//                ||r'||^2 (STEP_A2).  in = r, in2 = v, in3 = t, in4 = p, aux = r_hat.
//   kind 4 (s2): kind 1 with the partial sums t.s, t.t, r_hat.s, r_hat.t (DOT_B2, STEP_B2), aux = r_hat
constexpr int VK_NONE = 0, VK_S = 1, VK_P = 2, VK_R = 3, VK_S2 = 4, VK_PRO = 5;
struct VirtualIn {
    int kind;
    const double2 *in, *in2, *in3, *aux;
    double2* pout;
    const double2* in4 = nullptr;
    double2* rout = nullptr;
    double2* uio = nullptr;
};
// overlap: the 62-column overlapping-strip layout (VK_R / VK_S2 only; apply_tile<.., OV>): blocks[k].partials must then leave room for
// apply_block_nwg_overlap(ni, nj, rows) rows per block
hipError_t launch_apply_virtual(const ApplyBlock* blocks, int n, const EdgeRowsDev& e, const VirtualIn& V, const double2* xk, const double2* pq, double2* out,
                                double* edge_partials, const LazyScalars& scal, hipStream_t stream, bool overlap = false);
// b (unscaled) per perimeter row scattered into a dense vector that was zeroed by the caller; scaled!=0 writes D^-1 b
hipError_t launch_edge_rhs(const EdgeRowsDev& e, const double2* xk, const double2* pq, double2* rhs_out, int scaled,
                           double* partials /* [nwg*MAX_PARTIALS]: sum (D^-1 b)^2 x,y */, hipStream_t stream);

// ---- ordering between two queues without barrier packets: one-wave kernels.  signal: *counter += 1 once everything before it in
// its queue is complete and visible device-wide; wait: returns once *counter >= target (gives up after tens of seconds and sets *error).
hipError_t launch_queue_signal(uint32_t* counter, hipStream_t stream);
hipError_t launch_queue_wait(const uint32_t* counter, uint32_t target, uint32_t* error, hipStream_t stream, long long limit_ticks = QUEUE_WAIT_TICKS);
hipError_t launch_queue_signal_wait(uint32_t* counter, const uint32_t* other, uint32_t target, uint32_t* error, hipStream_t stream,
                                    long long limit_ticks = QUEUE_WAIT_TICKS);

// ---- the reference's ASSEMBLED system on the device (RowCompressedMatrixSystem2d values, smooth.zig:923-1113), for the introspection entry
// points tm_smoother_assemble_csr / tm_smoother_apply_reference_order: row_ptr = the CSR row pointers of the rank-local rows (host-built
// from the plan); interior rows get StencilData.init's nine values in ascending column order (smooth.zig:171-216, 923-992: stencil_coefs,
// the reference's expression order), perimeter rows their static coefficients or -- smoothed rows -- the same nine values by slot
hipError_t launch_assemble_interior(const double2* xk, const double2* pq, int ni, int nj, const int32_t* row_ptr /* of the block's node (0,0) */,
                                    double* vx, double* vy, hipStream_t stream);
hipError_t launch_assemble_edge(const EdgeRowsDev& e, const double2* xk, const double2* pq, const int32_t* row_ptr, double* vx, double* vy, hipStream_t stream);
// out = A in, the sum of a row's products in CSR order, un-fused and unscaled: BiCGStab.zig:424-435 bit for bit
hipError_t launch_csr_product(int64_t n, const int32_t* row_ptr, const int32_t* col, const double* vx, const double* vy, const double2* in, double2* out,
                              hipStream_t stream);
hipError_t launch_delay_us(double us, hipStream_t stream);   // measurement support: a one-wave kernel that lasts `us` microseconds

// ---- reductions: sum partial rows [nwg][MAX_PARTIALS] in fixed order into red[MAX_PARTIALS]
hipError_t launch_finalize(const double* partials, int nwg, double* red, hipStream_t stream);

// ---- K3: fused BiCGStab vector kernels with device-resident scalars
struct KrylovScalars {   // one per smoother, lives in device memory; index = component (0 x, 1 y)
    double rho[2], rho_old[2], alpha[2], omega[2], beta[2];
    double tol2[2];       // squared absolute tolerance on ||r||_2
    double rr[2];         // last ||r||^2 (or ||s||^2 on early exit)
    double rr0[2];        // ||r0||^2 of the current solve
    int32_t done[2];      // 0 active, 1 converged, 2 breakdown
    int32_t early[2];     // ||s|| <= tol in this iteration: omega := 0
    int32_t iters;        // iterations executed in the current solve
    int32_t _pad;
};
constexpr int VEC_BLOCK = 256;
int vec_nwg(int64_t n);
// step ids for launch_scalar_update: consume red[], update KrylovScalars
enum ScalarStep : int { STEP_INIT = 0, STEP_SIGMA = 1, STEP_SS = 2, STEP_TSTT = 3, STEP_RHO = 4, STEP_TOL = 5, STEP_SS_TSTT = 6, STEP_INIT2 = 7, STEP_A2 = 8, STEP_B2 = 9 };
// Two-kernel iteration (VK_R / VK_S2): rho comes from the second apply's reduction, rho' = r_hat.s - omega r_hat.t (= r_hat.r'), so
// the x / r update needs no reduction of its own and rides in front of the next first apply:
//   STEP_INIT2: STEP_INIT with alpha = omega = beta = 0 (nothing pending; p' = r)
//   STEP_A2:    red[0..1] = r_hat.v', red[2..3] = ||r'||^2 -> convergence / breakdown of the update just applied, alpha = rho / sigma
//   STEP_B2:    red[0..1] = t.s, red[2..3] = t.t, red[4..5] = r_hat.s, red[6..7] = r_hat.t -> omega, rho, beta
// STEP_SS_TSTT: red[4..5] = ||s||^2, red[0..1] = t.s, red[2..3] = t.t -- STEP_SS then STEP_TSTT from one reduction (DOT_IN_SS)
// STEP_TOL: red[0..1] = ||D^-1 b||^2 -> tol2 = max(atol, rtol*||D^-1 b||)^2
hipError_t launch_scalar_update(KrylovScalars* S, const double* red, int step, hipStream_t stream, double rtol = 0.0, double atol = 0.0);
hipError_t launch_finalize_scalar(const double* partials, int nwg, double* red, KrylovScalars* S, int step, hipStream_t stream, double rtol = 0.0,
                                  double atol = 0.0);
// Scalar steps without launches of their own (small meshes are bound by dependent launches, ~5 us each, and a BiCGStab iteration
// has four reductions): the kernel that CONSUMES the scalars applies the pending steps itself -- every workgroup sums the (few)
// partial rows in the same fixed order and runs the step on a private copy of the scalars; workgroup 0 publishes the copy to
// `S_out`, a different buffer than the `S_in` the other workgroups are still reading.  No device-wide fence is involved: kernel
// boundaries order everything.  nsteps = 0: plain read of S_in.
struct LazyStep {
    int step;                  // ScalarStep
    const double* partials;    // [nwg][MAX_PARTIALS]
    int nwg;
};
struct LazyScalars {
    const KrylovScalars* S_in = nullptr;
    KrylovScalars* S_out = nullptr;
    int nsteps = 0;
    LazyStep st[2] = {{0, nullptr, 0}, {0, nullptr, 0}};
};
inline LazyScalars plain_scalars(const KrylovScalars* S) {
    LazyScalars L;
    L.S_in = S;
    return L;
}
// p = r + beta (p - omega v)
hipError_t launch_p_update(const LazyScalars& S, const double2* r, double2* p, const double2* v, int64_t n, hipStream_t stream);
// s = r - alpha v ; partials: ||s||^2 (x,y)
hipError_t launch_s_update(const LazyScalars& S, const double2* r, const double2* v, double2* s, int64_t n, double* partials,
                           hipStream_t stream);
// u += alpha p + omega s ; r = s - omega t ; partials: r_hat.r (x,y), r.r (x,y)
// u += alpha*p_hat + omega*s_hat; r = s - omega*t (p_hat = p, s_hat = s without a preconditioner)
hipError_t launch_xr_update(const LazyScalars& S, double2* u, const double2* p_hat, const double2* s_hat, const double2* s, const double2* t, double2* r,
                            const double2* r_hat, int64_t n, double* partials, hipStream_t stream);
// the same with s = r - alpha v formed on the fly (s was never stored): reads p, r, v, t, r_hat, u; r is updated in place
hipError_t launch_xr_update_vs(const LazyScalars& S, double2* u, const double2* p, const double2* v, const double2* t, double2* r, const double2* r_hat,
                               int64_t n, double* partials, hipStream_t stream);
// ---- GMRES(m) (csrc/tm_gmres.hip; GMRES.zig:300-423): device-resident state of one solve, index = component (0 x, 1 y)
constexpr int GMRES_M = 30;   // restart length, GMRES.zig:21
struct GmresScalars {
    double H[2][(GMRES_M + 1) * GMRES_M];   // Hessenberg columns, h(row, col) at row + (m + 1) col (GMRES.zig:495-497), rotated in place
    double cs[2][GMRES_M], sn[2][GMRES_M];  // Givens rotations
    double g[2][GMRES_M + 1];               // rotated right-hand side; |g[j+1]| = the residual norm after column j
    double y[2][GMRES_M];                   // solution of the triangular system
    double tol[2], beta[2], resid[2], rr0[2], scale[2];
    double rtol_initial;
    int32_t tol_initial[2];
    int32_t cols_used[2];                   // columns of this cycle the component's update uses
    int32_t done[2];                        // resid <= tol (the component's solve is over)
    int32_t scale_skip[2];                  // see k_gm_divide
    int32_t j, cycle;
};
// one modified-Gram-Schmidt step in one pass: w -= red_prev[c] v_prev (and H[row_prev, j] = red_prev), partials = w . v_next (or ||w||^2
// when v_next == nullptr); v_prev == nullptr: the first step, nothing to subtract yet
hipError_t launch_gm_mgs(double2* w, const double2* v_prev, const double2* v_next, const double* red_prev, GmresScalars* G, int row_prev, int64_t n,
                         double* partials, hipStream_t stream);
hipError_t launch_gm_divide(double2* dst, const double2* src, const GmresScalars* G, int64_t n, hipStream_t stream);   // dst = src / G->scale[c]
hipError_t launch_gm_update(double2* u, const double2* V, int64_t ld, const GmresScalars* G, int64_t n, hipStream_t stream);   // u += V y
hipError_t launch_gm_tol(GmresScalars* G, const double* red, double rtol, double atol, hipStream_t stream);
hipError_t launch_gm_begin(GmresScalars* G, const double* red, hipStream_t stream);
hipError_t launch_gm_column(GmresScalars* G, const double* red, hipStream_t stream);
hipError_t launch_gm_backsub(GmresScalars* G, hipStream_t stream);

// K7: partials sum (xk-u)^2 (x,y); xk <- u
hipError_t launch_residual_copyback(double2* xk, const double2* u, int64_t n, double* partials, hipStream_t stream);
// gather rows for the halo exchange: dst[k] = src[ids[k]]
hipError_t launch_stream(int kind, double2* a, const double2* b, const double2* c, double s, int64_t n, hipStream_t stream);
hipError_t launch_gather_rows(const double2* src, const int32_t* ids, int64_t n, double2* dst, hipStream_t stream);

// out perimeter <- in perimeter of one block (fixed boundary of the stand-alone relax sweep)
hipError_t launch_copy_perimeter(const double2* in, double2* out, int ni, int nj, hipStream_t stream);
hipError_t launch_perimeter_sub(const double2* in, const double2* h, double2* out, int ni, int nj, hipStream_t stream);   // out = in - h + out on the block's perimeter (tm_gmres.hip)
// the perimeter values of f as Dirichlet data of a block's cycle: f_ring -= (D^-1 A)_ring,p f_p with the originals saved in `save` at the same
// positions, and their restoration (tm_gmres.hip)
hipError_t launch_ring_dirichlet(double2* f, const double2* X, const double2* PQ, double2* save, int ni, int nj, hipStream_t stream);
hipError_t launch_ring_restore(double2* f, const double2* save, int ni, int nj, hipStream_t stream);

// ---- K1 TFI
hipError_t launch_tfi_block(double2* xy, int ni, int nj, const double2* x_i_min, const double2* x_i_max, const double2* x_j_min,
                            const double2* x_j_max, const double* s1, const double* s2, const double* t1, const double* t2,
                            hipStream_t stream);
hipError_t launch_tfi_linear2d(double2* xy, int ni, int nj, const double2* e_i_min, const double2* e_i_max, const double2* e_j_min,
                               const double2* e_j_max, hipStream_t stream);

// ---- K6 white control function (wall_control_function.zig:70-473), blocks 0,1 + connection 0
struct WhiteArgs {
    const double2* x0;   // block 0 coordinates
    const double2* x1;   // block 1 coordinates
    double2* pq0;        // control function of block 0
    double2* pq1;
    int ni0, nj0, ni1, nj1;
    // leading-edge node across connection 0 (RangeFillMatrixIterator of connection 0, first point)
    int le_p0, le_p1, le_fi0, le_fi1, le_dir0;
    double ds_target, theta_target;
};
hipError_t launch_white(const WhiteArgs& w, int update, hipStream_t stream);

// ---- multigrid transfer kernels (one block, one pair of levels).  Coarse node c of a coarsened direction sits on fine node
// min(2c, n_fine - 1): standard vertex coarsening, with one short last cell when n_fine is even (4096 -> 2049 -> 1025 ...).
struct MgPair {
    int nif, njf, nic, njc;
    int ci, cj;   // direction coarsened between the two levels (1) or kept (0)
};
// coarse(ci,cj) = scale * fine(f(ci), f(cj)), every node incl. the perimeter (frozen coordinates: scale 1; control function: P x2
// where i is coarsened, Q x2 where j is -- a first-derivative coefficient in index space)
hipError_t launch_mg_inject(const double2* fine, double2* coarse, const MgPair& g, double scale_x, double scale_y, hipStream_t stream);
// f_c = (s_i s_j)^2 * full weighting (1/4 1/2 1/4 per coarsened direction) of the UNscaled fine residual, divided by the coarse
// diagonal (from the coarse coordinates X_coarse); interior coarse nodes, perimeter untouched (0)
hipError_t launch_mg_restrict(const double2* r_fine, const double2* X_coarse, double2* f_coarse, const MgPair& g, hipStream_t stream);
// First post-smoothing sweep of a level with the prolongation in front of it folded in: out = e' + omega (f - D^-1 A e'),
// e' = e_fine + bilinear interpolation of e_coarse formed as the rows enter K2's window (k_mg_prolong_add's expression: the same
// bits), e' itself never stored.  a.in = e_fine, a.in2 = e_coarse, a.aux = f, a.xk / a.pq = the level's frozen field.
hipError_t launch_mg_prolong_smooth(const ApplyBlock& a, const MgPair& g, hipStream_t stream);
// TWO chained operator applications of a level in one pass (k_mg_pair): kind 0 = two damped-Jacobi sweeps of D^-1 A e = f
// (in = e with zero perimeter, out = the result; coarse != nullptr: e = e + interpolation of the coarse correction first);
// kind 1 = the first two sweeps from e = 0 and the residual behind them (in = f, out = e2, out2 = a_ii (f - D^-1 A e2)).
// Interior nodes only are written.  Bit-identical to the K2 launches it replaces.
struct MgPairArgs {
    const double2* in = nullptr;
    const double2* xk = nullptr;
    const double2* pq = nullptr;
    const double2* f = nullptr;       // kind 0: the right-hand side (kind 1: `in` is the right-hand side)
    double2* out = nullptr;
    double2* out2 = nullptr;          // kind 1
    const double2* coarse = nullptr;  // kind 0 with the prolongation folded in
    // kind 1 with the restriction folded in (mg_pair_restrict_supported): the coarse level's frozen field and its right-hand side;
    // out2 is then not written
    const double2* xc = nullptr;
    double2* fc = nullptr;
    int nic = 0, njc = 0, ci = 0, cj = 0;
    int ni = 0, nj = 0;
    double omega = 1.0;
};
bool mg_pair_supported(int ni, int nj);
bool mg_pair_restrict_supported(int ni, int nj, int ci, int cj);
hipError_t launch_mg_pair(const MgPairArgs& a, int kind, hipStream_t stream);
// e_fine += bilinear interpolation of e_coarse, interior fine nodes
hipError_t launch_mg_prolong_add(const double2* e_coarse, double2* e_fine, const MgPair& g, hipStream_t stream);
// out = omega * f on interior nodes, 0 on the perimeter: the first damped-Jacobi sweep from a zero guess
hipError_t launch_mg_scale(const double2* f, double2* out, int ni, int nj, double omega, hipStream_t stream);

// ---- K8 export: interleaved (i*nj + j) block -> two planes with i fastest (cgns.zig:75-104)
hipError_t launch_soa_planes(const double2* in, double* plane0, double* plane1, int ni, int nj, hipStream_t stream);

// test hook: acos(x[i]) and atan2(y[i], x[i]) with the device build of tm_refmath.h (device pointers)
hipError_t launch_debug_white_math(const double* x, const double* y, uint64_t n, double* out_acos, double* out_atan2, hipStream_t stream);

}  // namespace tmh
