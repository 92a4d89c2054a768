// harness.cpp -- plays the role of the Zig caller (reference src/gui/main.zig:30-56, src/wasm/lib.zig:35-55):
// build edges -> Block2d.init (TFI on the MI355X) -> Mesh with connections -> smooth.mesh -> log like the reference.
//
//   tm_harness strip <nblocks> <ni> <nj> <iterations> [relax|bicgstab|mg] [until <tol>] [write <file.xyz>] [dump.bin]
//   tm_harness single <ni> <nj> <iterations> [relax|bicgstab|mg] [until <tol>] [write <file.xyz>] [dump.bin]
//   tm_harness csr                                     the linear-solver slot (seam 2) on the reference's 5 x 5 known answer
//   tm_harness ranks <world> <rank> <idfile> <nblocks> <ni> <nj> <iterations> [relax|bicgstab] [dump.bin]
//        one process per GPU (start each with HIP_VISIBLE_DEVICES=<its GPU>): the strip's blocks are dealt to the ranks in order, the
//        interface rows travel by the library's own RCCL transport (tm_rccl_*); rank 0 writes the ncclUniqueId to <idfile>, the
//        others wait for it (the rendezvous a Zig caller would do over MPI or a file); dump.bin = this rank's OWNED blocks
//
// The synthetic edges are those of SURVEY.md 8d (config 2 / config 4), identical to turbomesh_amd/configs.py.
// dump.bin (optional): all block coordinates as raw f64 after smoothing, for the parity test.
#include "turbomesh.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace core;
using discrete::Block2d;
using discrete::Edge;
using types::Vec2d;

static Edge uniformEdge(std::vector<Vec2d> pts) {
    Edge e;
    e.clustering = clustering::create(clustering::Uniform{}, pts.size());
    e.points = std::move(pts);
    return e;
}
static void snap(Block2d& b, const Edge& i_min, const Edge& i_max, const Edge& j_min, const Edge& j_max) {
    const std::size_t ni = b.points.size[0], nj = b.points.size[1];
    for (std::size_t i = 0; i < ni; ++i) {
        b.points.data[b.points.index(i, 0)] = i_min.points[i];
        b.points.data[b.points.index(i, nj - 1)] = i_max.points[i];
    }
    for (std::size_t j = 0; j < nj; ++j) {
        b.points.data[b.points.index(0, j)] = j_min.points[j];
        b.points.data[b.points.index(ni - 1, j)] = j_max.points[j];
    }
}

static discrete::Mesh buildStrip(std::size_t nblocks, std::size_t ni, std::size_t nj) {
    const double A = 0.1, two_pi = 2 * M_PI;
    const auto t = clustering::create(clustering::Uniform{}, nj);
    const auto s = clustering::create(clustering::Uniform{}, ni);
    auto curve = [&](std::size_t k) {
        std::vector<Vec2d> c(nj);
        for (std::size_t j = 0; j < nj; ++j) c[j] = Vec2d{{t[j], static_cast<double>(k) + A * (1.0 - 2.0 * static_cast<double>(k) / static_cast<double>(nblocks)) * std::sin(two_pi * t[j])}};
        c[0] = Vec2d{{0.0, static_cast<double>(k)}};
        c[nj - 1] = Vec2d{{1.0, static_cast<double>(k)}};
        return c;
    };
    discrete::Mesh mesh;
    for (std::size_t k = 0; k < nblocks; ++k) {
        const auto c0 = curve(k), c1 = curve(k + 1);
        std::vector<Vec2d> left(ni), right(ni);
        for (std::size_t i = 0; i < ni; ++i) {
            left[i] = Vec2d{{0.0, static_cast<double>(k) + s[i]}};
            right[i] = Vec2d{{1.0, static_cast<double>(k) + s[i]}};
        }
        left[0] = c0[0]; left[ni - 1] = c1[0];
        right[0] = c0[nj - 1]; right[ni - 1] = c1[nj - 1];
        const Edge i_min = uniformEdge(left), i_max = uniformEdge(right), j_min = uniformEdge(c0), j_max = uniformEdge(c1);
        Block2d b = Block2d::init(i_min, i_max, j_min, j_max);
        snap(b, i_min, i_max, j_min, j_max);
        mesh.addBlock("block_" + std::to_string(k), std::move(b));
    }
    for (std::size_t k = 0; k + 1 < nblocks; ++k)
        mesh.connections.push_back(boundary::Connection{{boundary::Range{k, boundary::Side::j_max, 0, nj - 1}, boundary::Range{k + 1, boundary::Side::j_min, 0, nj - 1}}, std::nullopt});
    return mesh;
}

static discrete::Mesh buildSingle(std::size_t ni, std::size_t nj) {
    const double A = 0.1, two_pi = 2 * M_PI;
    const auto s = clustering::create(clustering::Uniform{}, ni);
    const auto t = clustering::create(clustering::Uniform{}, nj);
    std::vector<Vec2d> lo(ni), up(ni), le(nj), ri(nj);
    for (std::size_t i = 0; i < ni; ++i) {
        lo[i] = Vec2d{{s[i], A * std::sin(two_pi * s[i])}};
        up[i] = Vec2d{{s[i], 1.0 - A * std::sin(two_pi * s[i])}};
    }
    for (std::size_t j = 0; j < nj; ++j) {
        le[j] = Vec2d{{0.0, t[j]}};
        ri[j] = Vec2d{{1.0, t[j]}};
    }
    lo[0] = Vec2d{{0, 0}}; lo[ni - 1] = Vec2d{{1, 0}};
    up[0] = Vec2d{{0, 1}}; up[ni - 1] = Vec2d{{1, 1}};
    const Edge i_min = uniformEdge(lo), i_max = uniformEdge(up), j_min = uniformEdge(le), j_max = uniformEdge(ri);
    discrete::Mesh mesh;
    Block2d b = Block2d::init(i_min, i_max, j_min, j_max);
    snap(b, i_min, i_max, j_min, j_max);
    mesh.addBlock("block", std::move(b));
    return mesh;
}

// Seam 2 from a compiled caller: a system assembled on the host -- here the reference's own known answer, umfpack.zig:71-97
// (A x = b, x = 1..5, given there in CSC; CSR below) as the x-system and 2 b as the y-system -- solved through tm_csr_solve.
static int csrKat() {
    const int32_t Ap[] = {0, 2, 5, 8, 9, 12};     // rows: [2 3 . . .] [3 . 4 . 6] [. -1 -3 2 .] [. . 1 . .] [. 4 2 . 1]
    const int32_t Ai[] = {0, 1, 0, 2, 4, 1, 2, 3, 2, 1, 2, 4};
    const double Ax[] = {2, 3, 3, 4, 6, -1, -3, 2, 1, 4, 2, 1};
    const double bx[] = {8, 45, -3, 3, 19};
    double by[5], x[5] = {0, 0, 0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) by[i] = 2 * bx[i];
    tm_solver_opt opt{};
    opt.tag = TM_SOLVER_HIP;
    opt.rtol = 1e-14;
    opt.max_inner = 200;
    opt.check_every = 1;
    tm_stats st{};
    const int rc = tm_csr_solve(5, Ap, Ai, Ax, nullptr, bx, by, x, y, &opt, &st);
    if (rc < 0) throw core::Error(rc, tm_last_error());
    double err = 0;
    for (int i = 0; i < 5; ++i) err = std::fmax(err, std::fmax(std::fabs(x[i] - (i + 1)), std::fabs(y[i] - 2 * (i + 1))));
    std::printf("csr kat: rc %d, inner iterations %llu, max error %.3e\n", rc, static_cast<unsigned long long>(st.inner_iterations), err);
    return (rc == 0 && err < 1e-9) ? 0 : 1;
}

// One rank of a multi-process run through the library's RCCL transport, from a compiled caller (what distributed.RcclHooks does
// from Python): unique id by file, communicator, hooks for this partition, handle, iterate, download the owned blocks.
static int runRank(int argc, char** argv) {
    if (argc < 9) throw Error(TM_E_ARG, "ranks <world> <rank> <idfile> <nblocks> <ni> <nj> <iterations> [relax|bicgstab] [dump.bin]");
    const int world = std::atoi(argv[2]), rank = std::atoi(argv[3]);
    const std::string idfile = argv[4];
    const std::size_t nb = std::strtoull(argv[5], nullptr, 10), ni = std::strtoull(argv[6], nullptr, 10), nj = std::strtoull(argv[7], nullptr, 10);
    const std::size_t iterations = std::strtoull(argv[8], nullptr, 10);
    if (world < 1 || rank < 0 || rank >= world || nb < static_cast<std::size_t>(world)) throw Error(TM_E_ARG, "ranks: need 0 <= rank < world <= nblocks");
    int a = 9;
    smoothing::solver::Option opt;
    if (a < argc && std::strcmp(argv[a], "relax") == 0) opt.inner = TM_INNER_RELAX;
    if (a < argc && (std::strcmp(argv[a], "relax") == 0 || std::strcmp(argv[a], "bicgstab") == 0)) ++a;
    opt.rtol = 1e-13;
    opt.max_inner = 5000;
    const char* librccl = std::getenv("TM_LIBRCCL");   // the copy the process should use; unset = default search
    unsigned char id[TM_RCCL_ID_BYTES];
    if (rank == 0) {
        check(tm_rccl_unique_id(librccl, id));
        const std::string tmp = idfile + ".tmp";
        std::FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) throw Error(TM_E_ARG, "cannot write " + tmp);
        std::fclose(f);
        if (std::rename(tmp.c_str(), idfile.c_str()) != 0) throw Error(TM_E_ARG, "cannot publish " + idfile);
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to a minute
            if (std::FILE* f = std::fopen(idfile.c_str(), "rb")) {
                got = std::fread(id, 1, sizeof(id), f) == sizeof(id);
                std::fclose(f);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!got) throw Error(TM_E_ARG, "no unique id appeared in " + idfile);
    }
    tm_rccl_comm* comm = nullptr;
    check(tm_rccl_comm_create(librccl, id, rank, world, &comm));
    discrete::Mesh mesh = buildStrip(nb, ni, nj);
    std::vector<int32_t> owner(nb);
    for (std::size_t b = 0; b < nb; ++b) owner[b] = static_cast<int32_t>(b * static_cast<std::size_t>(world) / nb);
    tm_stats st{};
    {
        smoothing::smooth::Desc d(mesh);   // the partition's exchange plan is derived from the same description the handle gets
        tm_comm_hooks hooks{};
        check(tm_rccl_hooks(comm, &d.desc, owner.data(), &hooks));
        smoothing::smooth::Smoother sm(mesh, opt, smoothing::wall_control_function::Algorithm::laplace(), &hooks);
        st = sm.iterate(iterations);
        sm.download();
    }
    tm_rccl_comm_destroy(comm);
    std::printf("rank %d of %d: outer %llu inner %llu not_converged %d last_residual %.17g\n", rank, world, static_cast<unsigned long long>(st.outer_iterations),
                static_cast<unsigned long long>(st.inner_iterations), st.not_converged, st.last_residual);
    if (a < argc) {
        std::FILE* f = std::fopen(argv[a], "wb");
        if (!f) throw Error(TM_E_ARG, "cannot open dump file");
        for (std::size_t b = 0; b < nb; ++b)
            if (owner[b] == rank) std::fwrite(mesh.blocks[b].points.data.data(), sizeof(Vec2d), mesh.blocks[b].points.data.size(), f);
        std::fclose(f);
    }
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc >= 2 && std::strcmp(argv[1], "csr") == 0) return csrKat();
        if (argc >= 2 && std::strcmp(argv[1], "ranks") == 0) return runRank(argc, argv);
        if (argc < 5) {
            std::fprintf(stderr, "usage: %s strip <nblocks> <ni> <nj> <iterations> [relax|bicgstab|mg] [until <tol>] [write <file.xyz>] [dump.bin]\n       %s single <ni> <nj> <iterations> [relax|bicgstab|mg] [until <tol>] [write <file.xyz>] [dump.bin]\n", argv[0], argv[0]);
            return 2;
        }
        int a = 2;
        discrete::Mesh mesh;
        if (std::strcmp(argv[1], "strip") == 0) {
            const std::size_t nb = std::strtoull(argv[a++], nullptr, 10), ni = std::strtoull(argv[a++], nullptr, 10), nj = std::strtoull(argv[a++], nullptr, 10);
            mesh = buildStrip(nb, ni, nj);
        } else {
            const std::size_t ni = std::strtoull(argv[a++], nullptr, 10), nj = std::strtoull(argv[a++], nullptr, 10);
            mesh = buildSingle(ni, nj);
        }
        const std::size_t iterations = std::strtoull(argv[a++], nullptr, 10);
        smoothing::solver::Option opt;
        if (a < argc && std::strcmp(argv[a], "relax") == 0) opt.inner = TM_INNER_RELAX;
        if (a < argc && std::strcmp(argv[a], "mg") == 0) opt.inner = TM_INNER_MG_BICGSTAB;
        if (a < argc && std::strcmp(argv[a], "gmres") == 0) opt.inner = TM_INNER_GMRES;   // GMRES(30) on the device (GMRES.zig:300-423)
        if (a < argc && std::strcmp(argv[a], "auto") == 0) opt.inner = TM_INNER_AUTO;
        const bool tight_default = a < argc && (std::strcmp(argv[a], "gmres") == 0 || std::strcmp(argv[a], "auto") == 0);
        if (a < argc && (std::strcmp(argv[a], "relax") == 0 || std::strcmp(argv[a], "bicgstab") == 0 || std::strcmp(argv[a], "mg") == 0 || tight_default)) ++a;
        opt.rtol = tight_default ? 0.0 : 1e-13;   // gmres / auto: the library's own default tolerance
        opt.max_inner = tight_default ? 0 : 5000;
        // "until <tol>" after the solver name: iterate to a residual through the device-resident handle instead of a fixed count
        double until = 0.0;
        std::string plot3d;
        if (a + 1 < argc && std::strcmp(argv[a], "until") == 0) {
            until = std::strtod(argv[a + 1], nullptr);
            a += 2;
        }
        if (a + 1 < argc && std::strcmp(argv[a], "write") == 0) {
            plot3d = argv[a + 1];
            a += 2;
        }
        // the reference's two log lines per outer iteration (smooth.zig:105, 136-137)
        tm_set_log([](void*, int32_t what, uint64_t n, double v) {
            if (what == 0) std::printf("info(smoothing): iteration: %llu\n", static_cast<unsigned long long>(n));
            else std::printf("info(smoothing): \tresidual: %.17g\n", v);
        }, nullptr);
        tm_stats st{};
        if (until > 0.0 || !plot3d.empty()) {
            smoothing::smooth::Smoother sm(mesh, opt, smoothing::wall_control_function::Algorithm::laplace());
            if (until > 0.0) {
                const bool reached = sm.iterateUntil(until, iterations, &st);
                std::printf("reached %d\n", reached ? 1 : 0);
            } else {
                st = sm.iterate(iterations);
            }
            sm.download();
            if (!plot3d.empty()) sm.write(plot3d);
        } else {
            st = smoothing::smooth::mesh(mesh, iterations, opt, smoothing::wall_control_function::Algorithm::laplace());
        }
        tm_set_log(nullptr, nullptr);
        std::printf("info(smoothing): elapsed time for smoothing: %.2f s\n", st.seconds);   // smooth.zig:159
        std::printf("inner_iterations %llu operator_sweeps %llu not_converged %d scaled_residual_rms %.3e\n", static_cast<unsigned long long>(st.inner_iterations),
                    static_cast<unsigned long long>(st.operator_sweeps), st.not_converged, st.scaled_residual_rms);
        if (a < argc) {
            std::FILE* f = std::fopen(argv[a], "wb");
            if (!f) throw Error(TM_E_ARG, "cannot open dump file");
            for (const auto& b : mesh.blocks) std::fwrite(b.points.data.data(), sizeof(Vec2d), b.points.data.size(), f);
            std::fclose(f);
        }
        return 0;
    } catch (const core::Error& e) {
        std::fprintf(stderr, "error(%d): %s\n", e.code, e.what());
        return 1;
    }
}
