"""CPU, world_size 2 and 3 over gloo: the N>1 path's host logic -- rank-local plans (tm_plan_local, no GPU) and the
torch.distributed halo exchange / all-reduce of turbomesh_amd.distributed.HaloExchanger (the same code the RCCL job runs,
on CPU tensors).  Checks:
  - every rank's ghost rows arrive with exactly the owner's values (pack order, peer offsets, p2p pairing);
  - the ghost set is sufficient: every column of every owned perimeter row is owned or ghost, so the row-wise
    mat-vec assembled from rank-local data equals the global oracle mat-vec (reference smooth.zig rows);
  - the partition is a partition (owned rows of all ranks = all rows, no overlap); all-reduce sums."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from turbomesh_amd import configs, distributed as tmd
from turbomesh_amd.smoothing import smooth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mesh(kind):
    if kind == "strip4":
        return configs.strip(4, 9, 12, tfi=oracle_tfi, reverse_odd=True)
    if kind == "two_by_two":
        return configs.two_by_two(8, 9, tfi=oracle_tfi)
    raise ValueError(kind)


def _worker(rank, world, port, kind, owner, q):
    try:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        mesh = _mesh(kind)
        om = OracleMesh(mesh)
        s = oracle.System(om)
        s.fill(0)
        dof = s.dof
        rng = np.random.default_rng(123)           # the same global vector on every rank
        v = rng.standard_normal((dof, 2))
        plan = tmd.local_plan(mesh, owner, rank, world)
        n_owned, n_ghost = plan["n_owned"], plan["n_ghost"]
        starts = np.cumsum([0] + [b.points.data.shape[0] * b.points.data.shape[1] for b in mesh.blocks])
        owned_gid = np.concatenate([np.arange(starts[b], starts[b + 1]) for b in plan["owned_blocks"]]) if len(plan["owned_blocks"]) else np.zeros(0, int)
        assert len(owned_gid) == n_owned
        local = np.full((n_owned + n_ghost, 2), np.nan)
        local[:n_owned] = v[owned_gid]
        # pack -> exchange -> ghosts
        send = torch.from_numpy(local[plan["send_ids"]].reshape(-1).copy()) if plan["n_send"] else torch.zeros(0, dtype=torch.float64)
        assert np.array_equal(owned_gid[plan["send_ids"]], plan["send_gid"])
        recv = torch.full((2 * n_ghost,), float("nan"), dtype=torch.float64)
        x = tmd.HaloExchanger(plan)
        x.exchange(send, recv)
        local[n_owned:] = recv.numpy().reshape(-1, 2)
        assert np.array_equal(local[n_owned:], v[plan["ghost_gid"]]), "ghost rows differ from the owner's values"
        # row-wise mat-vec of the owned PERIMETER rows from rank-local data only
        g2l = {int(g): k for k, g in enumerate(owned_gid)}
        g2l.update({int(g): n_owned + k for k, g in enumerate(plan["ghost_gid"])})
        rows = smooth.plan_rows(mesh)
        A = s.csr()
        ref = np.stack([A @ v[:, 0], A @ v[:, 1]], 1)
        mine = [k for k, g in enumerate(rows["row"]) if int(g) in g2l and g2l[int(g)] < n_owned]
        for k in mine:
            g = int(rows["row"][k])
            cols = rows["cols"][k, :rows["ncols"][k]]
            assert all(int(c) in g2l for c in cols), f"rank {rank}: row {g} reads a row that is neither owned nor ghost"
            r = A.getrow(g)
            acc = np.zeros(2)
            for c, a in zip(r.indices, r.data):
                acc = acc + a * local[g2l[int(c)]]
            assert np.allclose(acc, ref[g], rtol=1e-13, atol=1e-13)
        # depth-2 halo: every column of every ghost row this rank evaluates itself is owned or ghost as well
        for gid, cols in zip(plan["ghost_row_gid"], plan["ghost_row_cols"]):
            assert int(gid) in g2l and g2l[int(gid)] >= n_owned
            assert all(int(c) in g2l for c in cols if c >= 0), f"rank {rank}: ghost row {gid} reads a row that is neither owned nor ghost"
        # partition check + all-reduce
        t = torch.tensor([float(n_owned), float(len(mine)), 1.0, float(rank)], dtype=torch.float64)
        x.allreduce(t)
        assert int(t[0]) == dof and int(t[1]) == len(rows["row"]) and int(t[2]) == world and int(t[3]) == sum(range(world))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


@pytest.mark.parametrize("kind,world,owner", [
    ("strip4", 2, [0, 0, 1, 1]),
    ("strip4", 2, [0, 1, 0, 1]),          # interleaved ownership: every interface crosses ranks
    ("strip4", 3, [0, 1, 2, 2]),
    ("two_by_two", 2, [0, 1, 1, 0]),      # junction point shared by both ranks
])
def test_halo_exchange_and_plan_over_gloo(kind, world, owner):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, owner, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, res in sorted(results):
        assert res == "ok", f"rank {rank}:\n{res}"


def test_local_plan_single_rank_has_no_ghosts():
    mesh = _mesh("strip4")
    plan = tmd.local_plan(mesh, [0, 0, 0, 0], 0, 1)
    assert plan["n_ghost"] == 0 and plan["n_send"] == 0 and len(plan["peer_rank"]) == 0
    assert plan["n_owned"] == 4 * 9 * 12


def test_strip_for_rank_matches_full_strip():
    full = configs.strip(4, 9, 12, tfi=oracle_tfi)
    part = configs.strip(4, 9, 12, tfi=oracle_tfi, only_blocks={2})
    assert part.blocks[0].points.data is None and part.blocks[2].points.data is not None
    assert np.array_equal(part.blocks[2].points.data, full.blocks[2].points.data)   # interface curves are bitwise reproducible per rank


def test_direct_send_plans():
    # a strip's interfaces run along whole block rows: each peer's send list is ONE run of local rows, so a handle sends straight
    # from its vector; a 2 x 2 arrangement has column interfaces (strided rows) and keeps the pack kernel.  Depth-2 halo: the
    # solved side of an interface (ranges[0], the lower block) sends its interface row + first interior row -- what the upper block
    # needs to evaluate that `smoothed` interface row itself --, the slaved side (ranges[1]) its interface row + TWO interior rows
    # -- what the lower block needs to evaluate the upper block's first interior row (smooth.zig:1029-1032, 1071-1084)
    strip = configs.strip(4, 9, 12, tfi=oracle_tfi, reverse_odd=True)
    for rank in range(4):
        p = tmd.local_plan(strip, [0, 1, 2, 3], rank, 4)
        assert p["direct_send"] and len(p["peer_rank"]) == (1 if rank in (0, 3) else 2)
        for k in range(len(p["peer_rank"])):
            o, c = int(p["send_offset"][k]), int(p["send_count"][k])
            assert c == (3 * 12 if p["peer_rank"][k] < rank else 2 * 12)
            assert np.array_equal(p["send_ids"][o:o + c], p["send_first"][k] + np.arange(c))
        # the depth-1 part of the halo, with the rows' own definitions: the neighbour's first interior row (interior nodes + its two
        # wall nodes) seen from below, its interface row seen from above
        kinds = np.bincount(p["ghost_row_kind"], minlength=6)
        below, above = rank < 3, rank > 0
        assert kinds[5] == (10 if below else 0) and kinds[1] == (10 if above else 0) and kinds[0] == (2 if below else 0) + (2 if above else 0)
    grid = configs.two_by_two(8, 9, tfi=oracle_tfi)
    assert not any(tmd.local_plan(grid, [0, 1, 2, 3], r, 4)["direct_send"] for r in range(4))
    assert not tmd.local_plan(strip, [0, 0, 0, 0], 0, 1)["direct_send"]   # nobody to send to
