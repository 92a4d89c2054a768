// turbomesh.hpp -- C++ host-side mirror of turbomesh's `core` module for the hot path, on top of the C-ABI of
// libtm_hip.so (include/tm_hip.h).  The reference's host language is Zig (no toolchain in this image); this header
// keeps its module / type / function names, argument meaning and error behaviour so a caller reads like the Zig code:
//
//   core::types      Vec2d, Mat2d (NaN-initialised, index = j + size[1]*i)      reference src/core/types.zig:6-101
//   core::clustering Uniform, Roberts, SingleHyperbolicClustering, create       src/core/clustering.zig:9-116
//   core::geometry   Line                                                        src/core/geometry.zig:17-41
//   core::boundary   Side, Range, Connection, Condition                          src/core/boundary.zig:8-187
//   core::discrete   Edge, EdgeView, Block2d, Mesh                               src/core/discrete.zig:12-217
//   core::tfi        linear2dBoundaryBlendedControlFunction, linear2d            src/core/tfi.zig:19-208
//   core::smoothing  solver::Option (+ .hip), wall_control_function::Algorithm, smooth::mesh
//                                                                                src/core/smoothing/{solver,wall_control_function,smooth}.zig
// Zig error unions become C++ exceptions of type core::Error carrying the tm_error code.
#pragma once
#include "../../include/tm_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

namespace core {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc < 0) throw Error(rc, tm_last_error());
}

namespace types {
using Index = std::size_t;   // types.zig:6
using Float = double;        // types.zig:7
struct Vec2d {               // types.zig:16-27
    Float data[2];
    static Vec2d init(Float a, Float b) { return Vec2d{{a, b}}; }
};
inline bool eqlApprox(Vec2d a, Vec2d b, Float tol) { return std::fabs(a.data[0] - b.data[0]) <= tol && std::fabs(a.data[1] - b.data[1]) <= tol; }
struct Mat2d {               // types.zig:78-101
    Index size[2];
    std::vector<Vec2d> data;
    static Mat2d init(Index n0, Index n1) {
        const Float nan = std::numeric_limits<Float>::quiet_NaN();
        Mat2d m{{n0, n1}, std::vector<Vec2d>(n0 * n1, Vec2d{{nan, nan}})};
        return m;
    }
    Index index(Index i, Index j) const { return j + size[1] * i; }
    Vec2d getIndex(Index i, Index j) const { return data[index(i, j)]; }
};
}  // namespace types

namespace clustering {   // clustering.zig
struct Uniform {};
struct Roberts { double alpha, beta; };
struct SingleHyperbolicClustering { double delta_s; };
using Function = std::variant<Uniform, Roberts, SingleHyperbolicClustering>;
inline std::vector<double> create(const Function& f, types::Index n) {   // clustering.zig:110-116
    std::vector<double> u(n);
    if (std::holds_alternative<Uniform>(f)) {
        for (types::Index i = 0; i < n; ++i) u[i] = static_cast<double>(i) / static_cast<double>(n - 1);
    } else if (const auto* r = std::get_if<Roberts>(&f)) {
        for (types::Index i = 0; i < n; ++i) {
            const double x = static_cast<double>(i) / static_cast<double>(n - 1);
            const double tmp = std::pow((r->beta + 1.0) / (r->beta - 1.0), (x - r->alpha) / (1.0 - r->alpha));
            const double tbar = (r->beta + 2.0 * r->alpha) * tmp - r->beta + 2.0 * r->alpha;
            u[i] = tbar / ((2.0 * r->alpha + 1.0) * (1.0 + tmp));
        }
    } else {
        const auto& h = std::get<SingleHyperbolicClustering>(f);
        const double n_1 = static_cast<double>(n - 1), y = 1.0 / (n_1 * h.delta_s);
        double delta;
        if (y < 2.7829681) {
            const double yb = y - 1.0;
            delta = std::sqrt(6.0 * yb) * (1.0 + yb * (-0.15 + yb * (0.057321429 + yb * (-0.024907295 + yb * (0.0077424461 - 0.0010794123 * yb)))));
        } else {
            const double w = 1.0 / y - 0.028527431, v = std::log(y);
            delta = v + (1.0 + 1.0 / v) * std::log(2.0 * v) - 0.02041793 + w * (0.24902722 + w * (1.9496443 + w * (-2.6294547 + 8.56795911 * w)));
        }
        for (types::Index i = 0; i < n; ++i) u[i] = static_cast<double>(i) / n_1;
        for (types::Index i = 1; i < n; ++i) u[i] = 1.0 + std::tanh(0.5 * delta * (u[i] - 1.0)) / std::tanh(0.5 * delta);
    }
    return u;
}
}  // namespace clustering

namespace geometry {   // geometry.zig:17-41
struct Line {
    types::Vec2d start, end;
    std::vector<types::Vec2d> interpolate(const std::vector<double>& u) const {
        std::vector<types::Vec2d> out(u.size());
        const double dx = end.data[0] - start.data[0], dy = end.data[1] - start.data[1];
        for (std::size_t i = 0; i < u.size(); ++i) out[i] = types::Vec2d{{start.data[0] + u[i] * dx, start.data[1] + u[i] * dy}};
        return out;
    }
};
}  // namespace geometry

namespace boundary {   // boundary.zig
enum class Side : uint32_t { i_min = 0, i_max = 1, j_min = 2, j_max = 3 };
enum class ConditionTag : uint32_t { wall = 0, inlet = 1, outlet = 2 };
struct Range {
    std::size_t block;
    Side side;
    std::size_t start, end;
    std::size_t len() const { return start > end ? start - end + 1 : end - start + 1; }
};
struct Connection {
    Range ranges[2];
    std::optional<types::Vec2d> periodicity;   // ?Vec2d
};
struct Condition {
    Range range;
    ConditionTag kind;
};
}  // namespace boundary

namespace tfi {   // tfi.zig
// tfi.zig:112-122: same argument order; the reference's asserts come back as core::Error
inline void linear2dBoundaryBlendedControlFunction(types::Mat2d& data, const std::vector<types::Vec2d>& x_i_min, const std::vector<types::Vec2d>& x_i_max,
                                                   const std::vector<types::Vec2d>& x_j_min, const std::vector<types::Vec2d>& x_j_max,
                                                   const std::vector<double>& s1, const std::vector<double>& s2, const std::vector<double>& t1,
                                                   const std::vector<double>& t2) {
    const std::size_t n = x_i_min.size(), m = x_j_min.size();
    if (x_i_max.size() != n || s1.size() != n || s2.size() != n || x_j_max.size() != m || t1.size() != m || t2.size() != m || data.size[0] != n ||
        data.size[1] != m)
        throw Error(TM_E_SIZE, "edge / clustering / block sizes do not agree (tfi.zig:125-133)");
    check(tm_tfi_block(&data.data[0].data[0], n, m, &x_i_min[0].data[0], &x_i_max[0].data[0], &x_j_min[0].data[0], &x_j_max[0].data[0], s1.data(),
                       s2.data(), t1.data(), t2.data()));
}
inline void linear2d(std::vector<types::Vec2d>& data, const std::vector<types::Vec2d>& e_i_min, const std::vector<types::Vec2d>& e_i_max,
                     const std::vector<types::Vec2d>& e_j_min, const std::vector<types::Vec2d>& e_j_max) {   // tfi.zig:19-67
    const std::size_t n = e_i_min.size(), m = e_j_min.size();
    if (e_i_max.size() != n || e_j_max.size() != m) throw Error(TM_E_SIZE, "error.InconsistentSize (tfi.zig:30)");
    data.resize(n * m);
    check(tm_tfi_linear2d(&data[0].data[0], n, m, &e_i_min[0].data[0], &e_i_max[0].data[0], &e_j_min[0].data[0], &e_j_max[0].data[0]));
}
}  // namespace tfi

namespace discrete {   // discrete.zig
struct Edge {          // :12-36
    std::vector<types::Vec2d> points;
    std::vector<double> clustering;
    static Edge init(types::Index n, const geometry::Line& curve, const clustering::Function& cl) {
        Edge e;
        e.clustering = clustering::create(cl, n);
        e.points = curve.interpolate(e.clustering);
        return e;
    }
};
struct EdgeView {      // :94-136
    const Edge* edge;
    std::size_t start, end;
    std::size_t len() const { return start > end ? start - end + 1 : end - start + 1; }
};
inline Edge combine(const std::vector<EdgeView>& edges) {   // Edge.combine, :38-91
    const double tol = 1e-10;
    for (std::size_t i = 0; i + 1 < edges.size(); ++i)
        if (!types::eqlApprox(edges[i].edge->points[edges[i].end], edges[i + 1].edge->points[edges[i + 1].start], tol))
            throw Error(TM_E_MISMATCH, "edges cannot be combined as end points do not match");
    std::size_t n = 0;
    for (const auto& e : edges) n += e.len();
    n -= edges.size() - 1;
    Edge out;
    out.points.resize(n);
    out.clustering.resize(n);
    std::size_t pos = 0;
    for (const auto& e : edges) {
        const std::size_t len = e.len();
        for (std::size_t k = 0; k < len; ++k) out.points[pos + k] = e.edge->points[e.start > e.end ? e.start - k : e.start + k];
        pos += len - 1;
    }
    pos = 0;
    double last_value = 0.0;
    for (const auto& e : edges) {
        const std::size_t first = e.start > e.end ? e.end : e.start, last = e.start > e.end ? e.start : e.end;
        out.clustering[pos] = last_value;
        const double base = e.edge->clustering[first];
        std::size_t k = 1;
        for (std::size_t i = first + 1; i <= last; ++i, ++k) out.clustering[pos + k] = last_value + (e.edge->clustering[i] - base);
        pos += k - 1;
        last_value = out.clustering[pos];
    }
    for (auto& v : out.clustering) v /= last_value;
    return out;
}
struct Block2d {       // :138-164
    types::Mat2d points;
    static Block2d init(const Edge& i_min, const Edge& i_max, const Edge& j_min, const Edge& j_max) {
        Block2d b{types::Mat2d::init(i_min.points.size(), j_min.points.size())};
        tfi::linear2dBoundaryBlendedControlFunction(b.points, i_min.points, i_max.points, j_min.points, j_max.points, i_min.clustering, i_max.clustering,
                                                    j_min.clustering, j_max.clustering);
        return b;
    }
};
struct Mesh {          // :166-195
    std::vector<Block2d> blocks;
    std::vector<std::string> names;
    std::vector<boundary::Connection> connections;
    std::vector<boundary::Condition> boundary_conditions;
    void addBlock(const std::string& name, Block2d block) {
        blocks.push_back(std::move(block));
        names.push_back(name);
    }
};
}  // namespace discrete

namespace smoothing {
enum class Preconditioner { diagonal, ilu0 };   // preconditioner.zig
namespace solver {                              // solver.zig:10-27 + hip
enum class Tag : int32_t { gmres = 0, bicgstab = 1, umfpack = 2, petsc = 3, hip = 4 };
struct Option {
    Tag tag = Tag::hip;
    Preconditioner preconditioner = Preconditioner::diagonal;   // payload of gmres / bicgstab
    int32_t inner = TM_INNER_BICGSTAB;                          // payload of hip
    double rtol = 0, atol = 0, omega = 0;
    uint64_t max_inner = 0;
    uint32_t check_every = 0;
    bool single_sweep = false;
};
}  // namespace solver
namespace wall_control_function {               // wall_control_function.zig:10-20, 56-68
struct White { double ds_target; double theta_target = 1.5707963267948966; };
struct Algorithm {
    std::optional<White> white;   // nullopt = .laplace
    static Algorithm laplace() { return Algorithm{}; }
};
}  // namespace wall_control_function
namespace smooth {
namespace detail {
inline tm_range toRange(const boundary::Range& r) { return tm_range{r.block, static_cast<uint32_t>(r.side), 0, r.start, r.end}; }
inline void requirePlot3d(const std::string& filename) {   // discrete.zig:215 error.OutputFormatNotEnabled for what is not built in
    const auto dot = filename.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : filename.substr(dot);
    if (ext != ".xyz" && ext != ".p3d" && ext != ".x") throw Error(TM_E_UNSUPPORTED, "OutputFormatNotEnabled: " + ext + " (PLOT3D .xyz / .p3d / .x is built in; .cgns needs the cgns library)");
}
inline void writeHeader(std::FILE* f, const discrete::Mesh& m) {   // int32 nblocks | (int32 ni, nj) per block
    const int32_t nb = static_cast<int32_t>(m.blocks.size());
    std::fwrite(&nb, 4, 1, f);
    for (const auto& b : m.blocks) {
        const int32_t sz[2] = {static_cast<int32_t>(b.points.size[0]), static_cast<int32_t>(b.points.size[1])};
        std::fwrite(sz, 4, 2, f);
    }
}
}
// POD description of a Mesh for the C ABI (the arrays live as long as this object)
struct Desc {
    std::vector<tm_block> blocks;
    std::vector<tm_connection> conns;
    std::vector<tm_condition> bcs;
    tm_mesh_desc desc{};
    explicit Desc(discrete::Mesh& mesh_data) {
        for (auto& b : mesh_data.blocks) blocks.push_back(tm_block{&b.points.data[0].data[0], b.points.size[0], b.points.size[1]});
        for (const auto& c : mesh_data.connections) {
            tm_connection tc{};
            tc.r[0] = detail::toRange(c.ranges[0]);
            tc.r[1] = detail::toRange(c.ranges[1]);
            tc.has_periodicity = c.periodicity ? 1 : 0;
            if (c.periodicity) {
                tc.periodicity[0] = c.periodicity->data[0];
                tc.periodicity[1] = c.periodicity->data[1];
            }
            conns.push_back(tc);
        }
        for (const auto& b : mesh_data.boundary_conditions) bcs.push_back(tm_condition{detail::toRange(b.range), static_cast<uint32_t>(b.kind), 0});
        desc = tm_mesh_desc{blocks.data(), blocks.size(), conns.data(), conns.size(), bcs.data(), bcs.size()};
    }
};
inline tm_solver_opt toOpt(const solver::Option& o) {
    return tm_solver_opt{static_cast<int32_t>(o.tag), o.inner, o.rtol, o.atol, o.max_inner, o.check_every, o.single_sweep ? uint32_t{TM_OPT_SINGLE_SWEEP} : 0u, o.omega};
}
inline tm_control_fn toControl(const wall_control_function::Algorithm& a) {
    if (a.white) return tm_control_fn{TM_CF_WHITE, 0, a.white->ds_target, a.white->theta_target};
    return tm_control_fn{TM_CF_LAPLACE, 0, 0.0, 0.0};
}

// smooth.zig:74-80: mutates mesh_data.blocks[b].points.data in place
inline tm_stats mesh(discrete::Mesh& mesh_data, std::size_t iterations, const solver::Option& solver_option,
                     const wall_control_function::Algorithm& control_function_algorithm) {
    Desc d(mesh_data);
    tm_solver_opt so = toOpt(solver_option);
    tm_control_fn cf = toControl(control_function_algorithm);
    tm_stats st{};
    check(tm_smooth_mesh(&d.desc, iterations, &so, &cf, &st));
    return st;
}

// The same smoother with the mesh resident on the device between calls (tm_smoother_*): iterate a fixed count like the
// reference, or until the scaled nonlinear residual reaches a tolerance; write() = system.write (smooth.zig:396-414).
class Smoother {
   public:
    // hooks: the transport of a multi-process run (one process per GPU), e.g. filled by tm_rccl_hooks; nullptr = all blocks here
    Smoother(discrete::Mesh& mesh_data, const solver::Option& o, const wall_control_function::Algorithm& a, const tm_comm_hooks* hooks = nullptr)
        : mesh_(mesh_data), d_(mesh_data) {
        tm_solver_opt so = toOpt(o);
        tm_control_fn cf = toControl(a);
        check(tm_smoother_create(&d_.desc, &so, &cf, hooks, nullptr, &h_));
    }
    const tm_mesh_desc& desc() const { return d_.desc; }
    ~Smoother() { tm_smoother_destroy(h_); }
    Smoother(const Smoother&) = delete;
    Smoother& operator=(const Smoother&) = delete;
    tm_stats iterate(std::size_t iterations) {
        tm_stats st{};
        check(tm_smoother_iterate(h_, iterations, &st));
        return st;
    }
    bool iterateUntil(double scaled_residual_tol, std::size_t max_iterations, tm_stats* stats = nullptr) {
        tm_stats st{};
        const int rc = tm_smoother_iterate_until(h_, max_iterations, scaled_residual_tol, &st);
        check(rc);
        if (stats) *stats = st;
        return rc == TM_OK;
    }
    void download() { check(tm_smoother_download(h_, &d_.desc)); }
    // multi-block PLOT3D grid file (planes transposed on the device); `.cgns` needs the cgns library like the reference
    void write(const std::string& filename) {
        detail::requirePlot3d(filename);
        std::FILE* f = std::fopen(filename.c_str(), "wb");
        if (!f) throw Error(TM_E_ARG, "cannot open " + filename);
        detail::writeHeader(f, mesh_);
        std::vector<double> x, y;
        for (std::size_t b = 0; b < mesh_.blocks.size(); ++b) {
            const std::size_t n = mesh_.blocks[b].points.size[0] * mesh_.blocks[b].points.size[1];
            x.resize(n);
            y.resize(n);
            check(tm_smoother_export_soa(h_, b, x.data(), y.data(), nullptr, nullptr));
            std::fwrite(x.data(), sizeof(double), n, f);
            std::fwrite(y.data(), sizeof(double), n, f);
        }
        std::fclose(f);
    }

   private:
    discrete::Mesh& mesh_;
    Desc d_;
    tm_smoother* h_ = nullptr;
};
}  // namespace smooth
}  // namespace smoothing

namespace discrete {
// Mesh.write (discrete.zig:197-216): one plane per coordinate with i fastest (cgns.zig:75-104), transposed on the device
inline void write(const Mesh& m, const std::string& filename) {
    smoothing::smooth::detail::requirePlot3d(filename);
    std::FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) throw Error(TM_E_ARG, "cannot open " + filename);
    smoothing::smooth::detail::writeHeader(f, m);
    std::vector<double> x, y;
    for (const auto& b : m.blocks) {
        const std::size_t n = b.points.size[0] * b.points.size[1];
        x.resize(n);
        y.resize(n);
        check(tm_export_soa(&b.points.data[0].data[0], b.points.size[0], b.points.size[1], x.data(), y.data()));
        std::fwrite(x.data(), sizeof(double), n, f);
        std::fwrite(y.data(), sizeof(double), n, f);
    }
    std::fclose(f);
}
}  // namespace discrete
}  // namespace core
