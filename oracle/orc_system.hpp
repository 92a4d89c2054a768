// ORACLE (test infrastructure) -- RowCompressedMatrixSystem2d and friends.
// Follows reference src/core/smoothing/smooth.zig and wall_control_function.zig.
#pragma once
#include "orc_types.hpp"
#include "tm_oracle.h"

namespace orc {

// smooth.zig:1168-1174 (declaration order)
enum BlockBoundaryPointKind : int32_t { fixed = 0, smoothed = 1, connected = 2, laplacian_smoothed = 3, sliding_circ = 4 };

// smooth.zig:1334-1337
struct OverlappingPoint {
    Index global_id;
    Vec2d periodicity;
};
// smooth.zig:1219-1232
struct LaplacianPoint {
    std::vector<OverlappingPoint> overlapping_points;   // capacity 4 in the reference
    std::vector<int32_t> stencil_ids;                   // capacity 6 in the reference
    Vec2d rhs;
    Index globalId() const { return overlapping_points[0].global_id; }
};

// smooth.zig:1212-1529
struct BlockBoundaryPoints {
    std::vector<int32_t> kind;   // PointData(BlockBoundaryPointKind).buffer
    PointDataBufferIndexConverter index_converter;
    std::vector<LaplacianPoint> laplacian_points;
    void init(const IndexConverter& ic, const Mesh& mesh);
};

// wall_control_function.zig:10-68
struct White {
    Float ds_target;
    Float theta_target;
};
struct ControlFunction {
    std::vector<Vec2d> data;
    int algorithm = ORC_CF_LAPLACE;
    White white{0, 0};
    void init(Index dof, const Mesh& mesh, int algo, White w);
    void update(const Mesh& mesh);
};

// smooth.zig:171-216
struct StencilData {
    Float data[9];
    enum index { i_j = 0, ip1_j, im1_j, i_jp1, i_jm1, ip1_jp1, ip1_jm1, im1_jp1, im1_jm1 };
    Float get(index i) const { return data[i]; }
    static StencilData init(Vec2d im1_j, Vec2d ip1_j, Vec2d i_jm1, Vec2d i_jp1, Float P, Float Q);
};

// smooth.zig:277-1166
struct System {
    Mesh mesh;   // the reference holds *Mesh; block.pts alias the caller's arrays
    Index dof = 0;
    std::vector<int32_t> lhs_p, lhs_i;
    std::vector<Float> lhs_values, rhs_x, rhs_y, x_new, y_new;
    std::vector<Index> row_start;   // row_idx_range_start_for_each_block
    BlockBoundaryPoints boundary_points;
    ControlFunction control_function;
    IndexConverter index_converter;
    bool seeded_initial_guess = false;   // BiCGStab.zig:14 / GMRES.zig:16 (lives in the solver there)

    void init(const Mesh& m, int cf_algo, White w);
    void fill(Index iteration);
    void fillXSpecific();
    void fillYSpecific();
    void seedInitialGuess();
    Float commit(Float* dx2, Float* dy2);
    void matVec(const Float* x, Float* out) const;

   private:
    Index nzStart(Index row) const { return static_cast<Index>(lhs_p[row]); }
    void initNonZeroMatrixEntries();
    void initBoundaryData();
    void fillBlockInternalPointData();
    void fillBlockConnectionData();
    static void computeConnectionStencilPositions(const Connection& c, const RangeFillMatrixIterator& it, Index pos[9]);
};

void connectionDataCheck(const Mesh& mesh);   // smooth.zig:220-275

// solvers (orc_solvers.cpp)
struct SolveReport {
    uint64_t iters = 0;
    bool converged = true;
};
enum Precond { diagonal = 0, ilu0 = 1 };
struct CsrView {
    Index n;
    const int32_t* p;
    const int32_t* i;
    const Float* v;
};
SolveReport bicgstab(const CsrView& A, const Float* rhs, Float* x, Precond pc, Index max_iters, Float rtol, Float atol);
void ilu0_factor_apply(const CsrView& A, const Float* rhs, Float* lu_out, Float* out);   // BiCGStab.zig:178-277, 384-422
SolveReport gmres(const CsrView& A, const Float* rhs, Float* x, Precond pc, Index restart, Index max_iters, Float rtol,
                  Float atol);
SolveReport scaled_bicgstab(const CsrView& A, const Float* rhs, Float* x, Index max_iters, Float rtol, Float atol);
void banded_direct(const CsrView& A, const Float* rhs, Float* x);

}  // namespace orc
