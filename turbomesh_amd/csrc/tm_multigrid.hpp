// Block-local geometric multigrid for the frozen-coefficient error equation (SURVEY "next" N4): the preconditioner of
// TM_INNER_MG_BICGSTAB.  One V-cycle approximates e = (D^-1 A_II)^-1 f on the interior rows of ONE block with e = 0 on
// its perimeter; the coupling between blocks is left to the outer Krylov iteration, so the cycle itself needs no communication.
// (The perimeter unknowns are not left with their diagonal alone: the caller hands their values to the cycle as Dirichlet data -- the
// right-hand side of the first interior ring is f_I - (D^-1 A)_Ip f_p while the cycle runs -- and applies the perimeter rows to the interior
// corrections behind the cycles of all blocks, e_p = f_p - (D^-1 A)_pI e_I: Smoother::precondition.)
//   levels      vertex coarsening per direction while it has >= 5 nodes: coarse node c sits on fine node min(2c, n-1)
//               (4096 -> 2049 -> 1025 -> ... -> 3: one short last cell whenever n is even).  Point Jacobi only smooths along
//               strong couplings, so while the block's mean cell aspect ratio g11/g22 = |x_xi|^2/|x_eta|^2 is off by more than
//               4 only the strongly coupled direction is coarsened (semi-coarsening), until the levels are near-isotropic
//   operator    rediscretisation: the Winslow stencil (K2, MODE_MG_*) on the injected coordinates; row-equilibrated, so
//               levels need no h^2 factors; P,Q are first-derivative coefficients and double with the index spacing
//   smoother    damped Jacobi, omega = 0.8; the first pre-sweep starts from zero and is a plain scaling
//   transfers   full weighting / bilinear interpolation (tm_kernels: k_mg_restrict, k_mg_prolong_add)
#pragma once
#include "tm_kernels.h"
#include <vector>

namespace tmh {

class DeviceArena;

struct MgLevel {
    int ni = 0, nj = 0;
    int ci = 0, cj = 0;          // coarsened from the next finer level in i / j
    double2 *X = nullptr, *PQ = nullptr;     // frozen coordinates / control function of the level
    double2 *f = nullptr, *a = nullptr, *b = nullptr, *r = nullptr;   // rhs, two iterates (ping-pong), residual
};

class BlockMG {
   public:
    int nu_pre = 2, nu_post = 2, nu_coarsest = 8;
    int64_t fuse_prolong_min = 262144;   // nodes of a level from which the folded form is used
    bool fuse_prolong = true;   // the prolongation formed inside the first post-smoothing sweep (launch_mg_prolong_smooth); TM_MG_FUSE_PROLONG=0: k_mg_prolong_add
    double omega = 0.8;
    bool fuse_restrict = true;   // levels that fill the device: the full weighting rides behind the PRE pass's residual, which is never stored (TM_MG_RESTRICT_FUSED=0: off)
    bool use_pair = true;   // two operator applications per pass on every level of at least 5 x 5 nodes (k_mg_pair); TM_MG_PAIR=0: off
    // level 0 buffers are the caller's; coarser levels are carved from the arena
    // aspect = mean g11/g22 of the block's cells (1 = unknown / isotropic); worst_case = size the arena for any aspect
    void build(DeviceArena& arena, int ni, int nj, bool has_pq, double aspect, bool worst_case);
    static double aspect_of(const double* xy, int ni, int nj);   // host estimate from the caller's coordinates
    static double aspect_spread_of(const double* xy, int ni, int nj);   // standard deviation of log(g11/g22) over the same samples (0 = unknown / uniform)
    // refresh the level hierarchy from the fine frozen field (after every change of X / PQ)
    void set_field(const double2* X0, const double2* PQ0, hipStream_t stream);
    // z = V-cycle(f).  w0, w1: two fine scratch blocks whose perimeter is and stays zero.  z's perimeter is left zero.
    void vcycle(const double2* f, double2* z, double2* w0, double2* w1, hipStream_t stream);
    size_t nlevels() const { return L.size(); }

   private:
    std::vector<MgLevel> L;
    void smooth(const MgLevel& l, const double2* f, const double2* in, double2* out, hipStream_t st) const;
    void residual(const MgLevel& l, const double2* f, const double2* in, double2* out, hipStream_t st) const;
    MgPair pair(size_t fine) const;
};

}  // namespace tmh
