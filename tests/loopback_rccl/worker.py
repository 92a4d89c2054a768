#!/usr/bin/env python3
"""One rank of a several-ranks-on-ONE-GPU job that drives libtm_hip's own transport (csrc/tm_rccl.cpp) through the loopback librccl
(TEST INFRASTRUCTURE, tests/loopback_rccl/loopback_rccl.hip).  Started by tests/test_gpu_loopback_transport.py through
`python -m torch.distributed.run` with torch.distributed on gloo (rendezvous + gathering the result; no data-path role).

usage: worker.py <mode> <topology> <ni> <nj> <iterations> <out.json>
  mode      relax    -- Jacobi sweeps (pairs / triples schedule): every rank's blocks must equal the single-handle run bit for bit
            krylov   -- Picard + BiCGStab, hooked recurrence with ncclAllReduce: bit-identical to the torch.distributed (gloo) hooks at
                        world 2, and within 1e-10 rms of the single-handle run
            gmres    -- the same with GMRES(30) as the inner solver (every Gram-Schmidt inner product is an all-reduce)
            mg       -- the same with the multigrid-preconditioned BiCGStab (two more exchanges per preconditioner application: the perimeter rows
                        applied to the neighbours' corrections)
  topology  strip | strip_rev | junction   (strip: `world` x $TM_WORKER_BLOCKS_PER_RANK blocks stacked in i; junction: configs.two_by_two, one block per rank)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    mode, topology, ni, nj, its, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    bpr = int(os.environ.get("TM_WORKER_BLOCKS_PER_RANK", "1"))
    import numpy as np
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)   # every rank on the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from turbomesh_amd import configs, distributed as tmd
    from turbomesh_amd.smoothing import smooth, solver

    assert os.environ.get("TM_RCCL_LIB", "").endswith("libtm_loopback_rccl.so"), "this worker is for the loopback transport"

    def build(only=None):
        if topology == "junction":   # every rank builds all four blocks; only the owned ones are read by its handle
            assert world == 4 and bpr == 1
            return configs.two_by_two(ni, nj)
        nb = world * bpr
        kw = {"reverse_odd": True} if topology == "strip_rev" else {}
        return configs.strip(nb, ni, nj, only_blocks=only, **kw) if only is not None else configs.strip(nb, ni, nj, **kw)

    nb = 4 if topology == "junction" else world * bpr
    owner = [b // bpr for b in range(nb)]
    owned = [b for b in range(nb) if owner[b] == rank]
    opt = (solver.Option.hip(inner=solver.Inner.relax) if mode == "relax" else
           solver.Option.hip(inner=solver.Inner.gmres, rtol=1e-13, max_inner=20000) if mode == "gmres" else
           solver.Option.hip(inner=solver.Inner.mg_bicgstab, rtol=1e-13) if mode == "mg" else solver.Option.hip(rtol=1e-13))
    mesh = build(set(owned))
    h = tmd.RcclHooks(mesh, owner=owner, rank=rank, world=world, option=opt)
    st = h.iterate(its)
    h.smoother.download()
    h.close()
    result = {"rank": rank, "world": world, "outer_iterations": int(st["outer_iterations"]), "inner_iterations": int(st["inner_iterations"])}

    other = None
    if mode in ("krylov", "gmres", "mg"):   # the same job through the torch.distributed hooks (gloo: halo rows and scalars staged through the host)
        mesh_t = build(set(owned))
        ht = tmd.TorchHooks(mesh_t, owner=owner, rank=rank, world=world, option=opt)
        ht.iterate(its)
        ht.smoother.download()
        ht.smoother.close()
        other = all(np.array_equal(mesh.blocks[b].points.data, mesh_t.blocks[b].points.data) for b in owned)

    mine = torch.from_numpy(np.stack([mesh.blocks[b].points.data for b in owned]).copy())
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    flags = [None] * world
    dist.all_gather_object(flags, other)
    if rank == 0:
        whole = build()
        with smooth.Smoother(whole, opt) as ref:
            ref.iterate(its)
            ref.download()
        got = np.concatenate([p.numpy() for p in parts])
        want = np.stack([whole.blocks[b].points.data for b in range(nb)])
        result["bit_identical_to_single_handle"] = bool(np.array_equal(got, want))
        result["rms_vs_single_handle"] = float(np.sqrt(np.mean((got - want) ** 2)))
        result["bit_identical_to_torch_hooks"] = None if mode == "relax" else bool(all(flags))
        with open(out_path, "w") as f:
            json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
