// ORACLE (test infrastructure) -- CPU restatement of turbomesh's core data model.
// Follows reference src/core/types.zig, src/core/boundary.zig, and the index helpers of
// src/core/smoothing/smooth.zig.  Not part of the product; see tm_oracle.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cmath>
#include <string>
#include <vector>
#include <stdexcept>

namespace orc {

using Index = std::size_t;   // types.zig:6
using Float = double;        // types.zig:7

// types.zig:16-27
struct Vec2d {
    Float data[2];
};
inline Vec2d vinit(Float a, Float b) { return Vec2d{{a, b}}; }
// types.zig:47-75
inline Vec2d add(Vec2d a, Vec2d b) { return Vec2d{{a.data[0] + b.data[0], a.data[1] + b.data[1]}}; }
inline Vec2d sub(Vec2d a, Vec2d b) { return Vec2d{{a.data[0] - b.data[0], a.data[1] - b.data[1]}}; }
inline Vec2d scale(Float s, Vec2d v) { return Vec2d{{s * v.data[0], s * v.data[1]}}; }
inline Vec2d negate(Vec2d v) { return Vec2d{{-v.data[0], -v.data[1]}}; }
inline bool eqlApprox(Vec2d a, Vec2d b, Float tol) {   // types.zig:43-45
    return std::fabs(a.data[0] - b.data[0]) <= tol && std::fabs(a.data[1] - b.data[1]) <= tol;
}

// Error carrying a C-ABI code; thrown inside the oracle, mapped at the extern "C" boundary.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// discrete.zig:138-141 + types.zig:78-101: a block is a Mat2d of Vec2d, j fastest.
// The oracle never owns the coordinates: like smooth.mesh it mutates the caller's arrays.
struct Block {
    Index ni, nj;   // size[0], size[1]
    Vec2d* pts;
    Index index(Index i, Index j) const { return j + nj * i; }   // types.zig:94-96
    Index dof() const { return ni * nj; }
};

enum Side : uint32_t { i_min = 0, i_max = 1, j_min = 2, j_max = 3 };   // boundary.zig:8-13

struct Mesh;

// boundary.zig:100-117
struct RangeIterator {
    Index count = 0;
    bool has = false;
    Index idx = 0;
    std::ptrdiff_t increment = 0;
    bool next(Index& out) {
        if (!has) return false;
        out = idx;
        if (count > 0) {
            count -= 1;
            idx = static_cast<Index>(static_cast<std::ptrdiff_t>(idx) + increment);
        } else {
            has = false;
        }
        return true;
    }
};

// boundary.zig:15-98
struct Range {
    Index block;
    Side side;
    Index start, end;
    Index len() const { return start > end ? start - end + 1 : end - start + 1; }
    RangeIterator iterate(const Mesh& mesh) const;
    void endpoints(const Mesh& mesh, Index out[2]) const;
    int firstInternalPointShift(const Mesh& mesh) const;
};

// boundary.zig:119-162
struct Connection {
    Range ranges[2];
    bool has_periodicity;
    Vec2d periodicity;
    Index len() const { return ranges[0].len(); }
    Index lenInternal() const { return len() - 2; }
};

enum ConditionTag : uint32_t { wall = 0, inlet = 1, outlet = 2 };   // boundary.zig:178-182
struct Condition {
    Range range;
    ConditionTag kind;
};

// discrete.zig:166-171 (names dropped: not on the hot path)
struct Mesh {
    std::vector<Block> blocks;
    std::vector<Connection> connections;
    std::vector<Condition> boundary_conditions;
};

inline RangeIterator Range::iterate(const Mesh& mesh) const {   // boundary.zig:28-61
    const Block& b = mesh.blocks[block];
    RangeIterator it;
    it.has = true;
    switch (side) {
        case i_min: it.idx = b.index(start, 0); it.increment = static_cast<std::ptrdiff_t>(b.nj); break;
        case j_max: it.idx = b.index(b.ni - 1, start); it.increment = 1; break;
        case i_max: it.idx = b.index(start, b.nj - 1); it.increment = static_cast<std::ptrdiff_t>(b.nj); break;
        case j_min: it.idx = b.index(0, start); it.increment = 1; break;
    }
    if (start > end) {
        it.increment = -it.increment;
        it.count = start - end;
    } else {
        it.count = end - start;
    }
    return it;
}

inline void Range::endpoints(const Mesh& mesh, Index out[2]) const {   // boundary.zig:64-75
    const Block& b = mesh.blocks[block];
    switch (side) {
        case i_min: out[0] = start * b.nj; out[1] = end * b.nj; break;
        case j_max: { Index base = (b.ni - 1) * b.nj; out[0] = base + start; out[1] = base + end; break; }
        case i_max: out[0] = start * b.nj + b.nj - 1; out[1] = end * b.nj + b.nj - 1; break;
        case j_min: out[0] = start; out[1] = end; break;
    }
}

inline int Range::firstInternalPointShift(const Mesh& mesh) const {   // boundary.zig:78-97
    const Block& b = mesh.blocks[block];
    switch (side) {
        case i_min: return 1;
        case i_max: return -1;
        case j_min: return static_cast<int>(b.nj);
        case j_max: return -static_cast<int>(b.nj);
    }
    return 0;
}

// smooth.zig:1618-1668
struct IndexConverter {
    std::vector<Index> start;   // global_point_index_range_start
    const Mesh* mesh = nullptr;
    void init(const Mesh& m) {
        mesh = &m;
        start.resize(m.blocks.size());
        Index total = 0;
        for (Index b = 0; b < m.blocks.size(); ++b) {
            start[b] = total;
            total += m.blocks[b].dof();
        }
    }
    Index globalIndex(Index block, Index local) const { return start[block] + local; }
    void localIndex(Index global, Index& block, Index& local) const {
        Index b = start.size() - 1;
        while (global < start[b]) b -= 1;
        block = b;
        local = global - start[b];
    }
    void index2d(Index block, Index local, Index& i, Index& j) const {
        const Block& blk = mesh->blocks[block];
        i = local / blk.nj;
        j = local - i * blk.nj;
    }
};

// boundary.zig:219-285
struct PointDataBufferIndexConverter {
    std::vector<Index> block_range_start_table;
    const Mesh* mesh = nullptr;
    Index total = 0;
    void init(const Mesh& m) {
        mesh = &m;
        block_range_start_table.resize(m.blocks.size());
        total = 0;
        for (Index b = 0; b < m.blocks.size(); ++b) {
            block_range_start_table[b] = total;
            total += 2 * (m.blocks[b].nj + m.blocks[b].ni - 2);
        }
    }
    Index bufferIndex(Index block, Index i, Index j) const {
        const Block& b = mesh->blocks[block];
        Index k;
        if (i == 0) k = j;                                              // j_min
        else if (i == b.ni - 1) k = b.nj + 2 * (b.ni - 2) + j;          // j_max
        else if (j == 0) k = b.nj + (i - 1) * 2;                        // i_min
        else if (j == b.nj - 1) k = b.nj - 1 + i * 2;                   // i_max
        else throw Error(-2, "NotBoundaryIndex");                      // boundary.zig:280
        return block_range_start_table[block] + k;
    }
};

// smooth.zig:1531-1599
struct RangeFillMatrixIterator {
    Index count = 0;
    int first_internal_point_shift[2] = {0, 0};
    int in_connection_direction_shift[2] = {0, 0};
    Index position[2] = {0, 0};

    bool next(Index out[2]) {
        if (count == 0) return false;
        out[0] = position[0];
        out[1] = position[1];
        count -= 1;
        position[0] = static_cast<Index>(static_cast<std::ptrdiff_t>(position[0]) + in_connection_direction_shift[0]);
        position[1] = static_cast<Index>(static_cast<std::ptrdiff_t>(position[1]) + in_connection_direction_shift[1]);
        return true;
    }
    void limitToRangeInternalPoints() {   // smooth.zig:1551-1554
        Index tmp[2];
        next(tmp);
        count -= 1;
    }
    static RangeFillMatrixIterator init(const Connection& c, const Mesh& mesh) {
        RangeFillMatrixIterator d;
        for (int s = 0; s < 2; ++s) {
            const Block& b = mesh.blocks[c.ranges[s].block];
            Index start = c.ranges[s].start, end = c.ranges[s].end;
            switch (c.ranges[s].side) {
                case i_min:
                    d.first_internal_point_shift[s] = 1;
                    d.in_connection_direction_shift[s] = static_cast<int>(b.nj);
                    d.position[s] = b.index(start, 0);
                    break;
                case i_max:
                    d.first_internal_point_shift[s] = -1;
                    d.in_connection_direction_shift[s] = static_cast<int>(b.nj);
                    d.position[s] = b.index(start, b.nj - 1);
                    break;
                case j_min:
                    d.first_internal_point_shift[s] = static_cast<int>(b.nj);
                    d.in_connection_direction_shift[s] = 1;
                    d.position[s] = b.index(0, start);
                    break;
                case j_max:
                    d.first_internal_point_shift[s] = -static_cast<int>(b.nj);
                    d.in_connection_direction_shift[s] = 1;
                    d.position[s] = b.index(b.ni - 1, start);
                    break;
            }
            if (start > end) {
                d.in_connection_direction_shift[s] = -d.in_connection_direction_shift[s];
                d.count = start - end + 1;
            } else {
                d.count = end - start + 1;
            }
        }
        return d;
    }
};

}  // namespace orc
