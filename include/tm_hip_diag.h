/*
 * tm_hip_diag.h -- measurement and diagnostic entry points of libtm_hip.so.  NOT part of the drop-in surface (include/tm_hip.h): nothing
 * here replaces a seam of the reference, a Zig binding needs none of it.  bench.py and the tests use them; they are exported from the
 * same library so that what is measured is the product build.
 */
#ifndef TM_HIP_DIAG_H
#define TM_HIP_DIAG_H

#include "tm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement support (bench.py roofline): with every = k > 0, every k-th launch of the dominant kernel (K2
 * `winslow_apply` / K2x2 / K2x3) is bracketed by a pair of HIP events recorded on the handle's stream (0 = off); read returns
 * the summed elapsed milliseconds of the bracketed launches, how many were bracketed and how many ran since the last
 * read (and resets all three). */
int tm_smoother_profile(tm_smoother* s, int every);
int tm_smoother_profile_read(tm_smoother* s, double* k2_ms_total, uint64_t* k2_launches_timed, uint64_t* k2_launches);

/* Diagnostic: the STREAM-style ceiling of this GPU at a footprint of `bytes` per array (SURVEY 8d asks for it beside the 8 TB/s
 * specification): copy (b = a: 2 x bytes moved) and triad (a = b + s c: 3 x bytes), 16 B per lane, non-temporal loads and stores --
 * the access pattern of the library's vector kernels -- averaged over `iters` launches after 3 warm-up launches; GB/s = 1e9 B/s. */
int tm_stream_probe(uint64_t bytes, int32_t iters, double* copy_GBps, double* triad_GBps);

/* Diagnostic: acos(x[i]) and atan2(y[i], x[i]) exactly as the White kernels evaluate them on the device (csrc/tm_refmath.h: the
 * reference's libm algorithm, Zig std.math = musl's, wall_control_function.zig:298-308).  Host arrays in and out. */
int tm_white_math_probe(const double* x, const double* y, uint64_t n, double* out_acos, double* out_atan2);

/* Diagnostic: ILU(0) of one CSR matrix on the device as tm_csr_solve's TM_OPT_PRECOND_ILU0 builds it -- the factor in A's own pattern into
 * lu_out [nnz] and, when rhs / z_out are given, M^-1 rhs into z_out [n] (BiCGStab.zig:178-277, 384-422).  Host arrays. */
int tm_csr_ilu0_probe(uint64_t n, const int32_t* Ap, const int32_t* Ai, const double* Ax, const double* rhs /* may be NULL */, double* lu_out,
                      double* z_out /* may be NULL */);

/* How the handle orders the two queues of a pipelined pass (interior pass on its stream, perimeter-row chain + halo exchange on a second,
 * high-priority stream of its own): counters in device memory with one-wave announce / wait kernels need the two streams on different
 * hardware queues, which HIP does not promise -- so the handle TESTS it once, when the second stream is created (one announce-and-wait
 * round in both directions, limit 5 ms), and falls back to hipEvent ordering when the round does not complete.
 *   -1  no two-queue schedule has run on this handle (yet)
 *    0  counters (the self-test passed)
 *    1  events: TM_PAIR_SYNC=events in the environment when the handle was created
 *    2  events: several multi-rank handles live in this process (they could block each other through shared queues)
 *    3  events: the self-test found both streams on one hardware queue */
int tm_smoother_queue_ordering(const tm_smoother* s);

#ifdef __cplusplus
}
#endif
#endif /* TM_HIP_DIAG_H */
