// Host-side planning of the smoother (pure C++, no HIP): turns the mesh TOPOLOGY (block sizes,
// connections, boundary conditions) into the tables the device kernels consume:
//   - the perimeter-row table: kind, columns and coefficients of every non-interior row of the
//     global system (what the reference builds as CSR rows in smooth.zig:421-921);
//   - for a multi-GPU job, the rank-local numbering (owned rows, ghost rows) and the halo
//     exchange lists.
// Interior rows never appear here: K2 applies them matrix-free.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace tmh {

struct PlanError : std::runtime_error {
    int code;
    PlanError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

enum Side : uint32_t { SIDE_I_MIN = 0, SIDE_I_MAX = 1, SIDE_J_MIN = 2, SIDE_J_MAX = 3 };
// BlockBoundaryPointKind, smooth.zig:1168-1174
enum RowKind : int8_t {
    KIND_FIXED = 0, KIND_SMOOTHED = 1, KIND_CONNECTED = 2, KIND_JUNCTION = 3, KIND_SLIDING = 4,
    // not a reference kind: an INTERIOR node of a remote block (matrix-free 9-point row, K2's arithmetic) that a rank evaluates
    // redundantly as a ghost row (LocalPlan::ghost_rows); col[] = (i-1,j-1) (i-1,j) (i-1,j+1) (i,j-1) (i,j) (i,j+1) (i+1,j-1) (i+1,j) (i+1,j+1)
    KIND_INTERIOR = 5
};

struct TopoRange {
    int64_t block;
    uint32_t side;
    int64_t start, end;
};
struct TopoConn {
    TopoRange r[2];
    bool periodic;
    double per[2];
};
struct TopoCond {
    TopoRange range;
    uint32_t kind;
};
struct Topology {
    std::vector<int64_t> ni, nj;
    std::vector<TopoConn> conns;
    std::vector<TopoCond> bcs;
    std::vector<int64_t> start;   // global row id of node (0,0) of each block
    int64_t dof = 0;
    void finalize();              // fills start/dof, validates ranges
    int64_t nblocks() const { return static_cast<int64_t>(ni.size()); }
};

// One perimeter row of the global system, columns in ascending global id.
struct PlanRow {
    int64_t gid;
    int8_t kind;
    int8_t ncols;
    int8_t self;            // position of gid within col[]
    int64_t col[9];
    double cx[9], cy[9];    // static coefficients of the x / y system (smoothed rows: unused)
    int8_t slot[9];         // smoothed rows: StencilData index feeding each column
    int64_t metric[4];      // smoothed rows: global ids of im1_j, ip1_j, i_jm1, i_jp1
    double per[2];          // periodicity of the connection the row belongs to
    uint8_t flags;          // bit0 periodic, bit1 swap (Q,P)
    double rhs[2];          // static rhs
    uint8_t rhs_coord;      // bit0: rhs_x = own x coordinate at create time, bit1: rhs_y = own y
};

// Connection iterator data (RangeFillMatrixIterator, smooth.zig:1556-1598)
struct ConnShifts {
    int64_t count;
    int64_t first_internal[2];
    int64_t direction[2];
    int64_t position[2];   // block-local flat index of the first point on each side
};
ConnShifts conn_shifts(const Topology& t, const TopoConn& c);

// All perimeter rows of the mesh, ascending gid.  Throws PlanError on invalid topologies.
std::vector<PlanRow> build_rows(const Topology& t);

// Rank-local view for one process of a multi-GPU job (owner[b] = rank owning block b).
struct LocalPlan {
    int rank = 0, nranks = 1;
    std::vector<int64_t> owned_blocks;      // global block ids, ascending
    std::vector<int64_t> local_start;       // local vector index of node (0,0) of each owned block
    int64_t n_owned = 0;                    // owned rows
    std::vector<int64_t> ghost_gid;         // ghost rows (sorted by owner rank, then gid); local id = n_owned + k
    std::vector<PlanRow> rows;              // owned perimeter rows (global ids inside)
    // Depth-2 halo.  The remote rows my perimeter rows read (the depth-1 ghost set) can be EVALUATED here, one sweep ahead, from the
    // previous field: ghost_rows holds their definitions (perimeter rows of the neighbour as they stand in its table, or
    // KIND_INTERIOR rows for its first-interior nodes), and the ghost set above also contains every remote row THOSE read.  A pair of
    // relaxation sweeps then needs ONE exchange: X^k of the depth-2 set travels, the depth-1 rows of X^(k+1) are recomputed.
    std::vector<PlanRow> ghost_rows;
    // Depth-3 halo (triple_halo): ghost_rows2 = the definitions of every row of the depth-2 set; sweep triples evaluate THEM at level 1
    // and ghost_rows at level 2, and the ghost set holds every remote row those read (X^k of it travels once per triple)
    bool triple_halo = false;
    std::vector<PlanRow> ghost_rows2;
    // halo exchange: for peer k, my rows send_ids[send_off[k] .. +send_cnt[k]) (local ids) go to peer_rank[k];
    // its rows land in my ghost segment at [recv_off[k], +recv_cnt[k])
    std::vector<int32_t> peer_rank;
    std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;
    std::vector<int32_t> send_ids;
    // every peer's rows form ONE ascending run of local ids (true for interfaces along whole block rows, e.g. a strip of blocks): the
    // exchange then sends straight from the vector (send_first[k] = first local id of peer k's run) and no pack kernel is needed
    bool direct_send = false;
    std::vector<int64_t> send_first;
    std::unordered_map<int64_t, int64_t> ghost_index;   // gid -> position in ghost_gid
    int64_t to_local(int64_t gid) const;    // -1 if neither owned nor ghost
    const Topology* topo = nullptr;
};
bool triple_halo_for(const Topology& t, int nranks);
LocalPlan build_local_plan(const Topology& t, const std::vector<PlanRow>& all_rows, const std::vector<int32_t>& owner, int rank,
                           int nranks, bool allow_triples = true);

}  // namespace tmh
