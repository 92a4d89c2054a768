// Helpers shared by the translation units that implement the extern "C" surface (tm_api.cpp, tm_rccl.cpp).
#pragma once
#include "../../include/tm_hip_diag.h"
#include "tm_smoother.hpp"

#include <cmath>
#include <new>
#include <string>

namespace tmh {

inline thread_local std::string g_last_error;

template <class F>
inline int guarded(F&& f) {
    try {
        return f();
    } catch (const TmError& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const PlanError& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of host memory";
        return TM_E_MEMORY;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return TM_E_ARG;
    }
}
#ifndef HIPCHK
#define HIPCHK(x) hip_check((x), #x)
#endif

inline Topology topo_of(const tm_mesh_desc* mesh) {
    Topology t;
    if (!mesh || !mesh->blocks || mesh->nblocks == 0) throw TmError(TM_E_ARG, "mesh description without blocks");
    for (uint64_t b = 0; b < mesh->nblocks; ++b) {
        t.ni.push_back(static_cast<int64_t>(mesh->blocks[b].ni));
        t.nj.push_back(static_cast<int64_t>(mesh->blocks[b].nj));
    }
    auto rng = [](const tm_range& r) {
        return TopoRange{static_cast<int64_t>(r.block), r.side, static_cast<int64_t>(r.start), static_cast<int64_t>(r.end)};
    };
    for (uint64_t c = 0; c < mesh->nconns; ++c) {
        TopoConn tc;
        tc.r[0] = rng(mesh->conns[c].r[0]);
        tc.r[1] = rng(mesh->conns[c].r[1]);
        tc.periodic = mesh->conns[c].has_periodicity != 0;
        tc.per[0] = mesh->conns[c].periodicity[0];
        tc.per[1] = mesh->conns[c].periodicity[1];
        t.conns.push_back(tc);
    }
    for (uint64_t c = 0; c < mesh->nbcs; ++c) t.bcs.push_back(TopoCond{rng(mesh->bcs[c].range), mesh->bcs[c].kind});
    t.finalize();
    return t;
}


}  // namespace tmh
