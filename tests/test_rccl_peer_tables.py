"""CPU, no communicator: the send / receive tables `tm_rccl_hooks` hands to ncclSend / ncclRecv, for EVERY rank of a partition.

RCCL point-to-point inside ncclGroupStart / ncclGroupEnd hangs (it does not fail) when two ranks disagree about a transfer:
a send without its receive, or counts that differ.  With one GPU per test box the transport has only ever had its own rank as peer
(tests/test_gpu_rccl_transport.py), so the pairwise agreement is checked here on the tables themselves -- `tm_rccl_peer_table_build`
is the very function `tm_rccl_hooks` fills its communicator from (csrc/tm_rccl.cpp: peer_table_of).  Cases: BASELINE configs[3]'s
8 x 2048^2 strip over 2, 3 and 8 ranks (block partitions and an interleaved one), reversed interfaces, and the 2 x 2 junction mesh
(column interfaces: packed sends; a junction row shared by all four blocks, smooth.zig:1340-1514)."""
import ctypes as C

import numpy as np
import pytest

from tests.conftest import oracle_tfi
from turbomesh_amd import _capi, configs
from turbomesh_amd import distributed as tmd


def _table(mesh, owner, rank, world):
    md = _capi.MeshDesc(mesh, with_coordinates=False)
    t = _capi.tm_rccl_peer_table()
    own = (C.c_int32 * len(owner))(*owner)
    _capi.check(_capi.lib().tm_rccl_peer_table_build(md.ref(), own, rank, world, C.byref(t)))
    try:
        n = t.npeers
        get = lambda p: [int(p[k]) for k in range(n)]
        return {"peer": get(t.peer), "send_off": get(t.send_off), "send_cnt": get(t.send_cnt), "recv_off": get(t.recv_off),
                "recv_cnt": get(t.recv_cnt), "send_rows": int(t.send_rows), "recv_rows": int(t.recv_rows), "direct": bool(t.direct_send)}
    finally:
        _capi.lib().tm_rccl_peer_table_free(C.byref(t))


def _strip8(n=2048, reverse_odd=False):
    return configs.strip(8, n, n, only_blocks=set(), reverse_odd=reverse_odd)   # sizes + topology only: planning reads no coordinates


CASES = [
    ("strip8_2048_world2", lambda: _strip8(), 2, [0, 0, 0, 0, 1, 1, 1, 1]),
    ("strip8_2048_world3", lambda: _strip8(), 3, [0, 0, 0, 1, 1, 1, 2, 2]),
    ("strip8_2048_world4", lambda: _strip8(), 4, [0, 0, 1, 1, 2, 2, 3, 3]),
    ("strip8_2048_world8", lambda: _strip8(), 8, list(range(8))),
    ("strip8_2048_world8_reversed", lambda: _strip8(reverse_odd=True), 8, list(range(8))),
    ("strip8_2048_world3_interleaved", lambda: _strip8(), 3, [0, 1, 2, 0, 1, 2, 0, 1]),
    ("strip8_4096_world8_triple_halo", lambda: configs.strip(8, 4096, 4096, only_blocks=set()), 8, list(range(8))),
    ("strip4_4096_world2_triple_halo", lambda: configs.strip(4, 4096, 4096, only_blocks=set(), reverse_odd=True), 2, [0, 0, 1, 1]),
    ("two_by_two_world2", lambda: configs.two_by_two(8, 9, tfi=oracle_tfi), 2, [0, 1, 1, 0]),
    ("two_by_two_world3", lambda: configs.two_by_two(8, 9, tfi=oracle_tfi), 3, [0, 1, 2, 0]),
    ("two_by_two_world4", lambda: configs.two_by_two(8, 9, tfi=oracle_tfi), 4, [0, 1, 2, 3]),
]


@pytest.mark.parametrize("name,build,world,owner", CASES, ids=[c[0] for c in CASES])
def test_every_send_has_its_receive(name, build, world, owner):
    mesh = build()
    tabs = [_table(mesh, owner, r, world) for r in range(world)]
    plans = [tmd.local_plan(mesh, owner, r, world) for r in range(world)]
    transfers = 0
    for a, (t, p) in enumerate(zip(tabs, plans)):
        assert t["peer"] == sorted(set(t["peer"])) and a not in t["peer"]
        assert t["recv_rows"] == p["n_ghost"]
        assert t["send_rows"] == (p["n_owned"] + p["n_ghost"] if t["direct"] else p["n_send"])
        got = np.zeros(t["recv_rows"], dtype=np.int32)
        for k, b in enumerate(t["peer"]):
            assert 0 <= b < world
            u = tabs[b]
            assert a in u["peer"], f"rank {a} lists {b} as a peer, {b} does not list {a}"
            j = u["peer"].index(a)
            # the property whose violation hangs ncclGroupEnd: what a sends to b is what b receives from a, and vice versa
            assert t["send_cnt"][k] == u["recv_cnt"][j], (a, b)
            assert t["recv_cnt"][k] == u["send_cnt"][j], (a, b)
            assert t["send_cnt"][k] > 0 or t["recv_cnt"][k] > 0
            # offsets stay inside the buffers the hooks are called with
            assert 0 <= t["send_off"][k] and t["send_off"][k] + t["send_cnt"][k] <= t["send_rows"]
            assert 0 <= t["recv_off"][k] and t["recv_off"][k] + t["recv_cnt"][k] <= t["recv_rows"]
            if t["direct"]:   # sends straight from the vector: only rows this rank owns
                assert t["send_off"][k] + t["send_cnt"][k] <= p["n_owned"]
            got[t["recv_off"][k]:t["recv_off"][k] + t["recv_cnt"][k]] += 1
            # and the rows mean the same thing on both sides: b's send list towards a, in order, is a's ghost segment from b
            q = plans[b]
            so, sc = int(q["send_offset"][j]), int(q["send_count"][j])
            ro = t["recv_off"][k]
            assert np.array_equal(q["send_gid"][so:so + sc], p["ghost_gid"][ro:ro + sc]), (a, b)
            transfers += 1
        assert (got == 1).all(), f"rank {a}: ghost rows not covered exactly once by the receives"
    assert transfers > 0


def test_config4_message_sizes(monkeypatch):
    # BASELINE configs[3], one block per GPU.  Blocks of half a million nodes and more take sweep TRIPLES: a depth-3 halo, one exchange
    # per triple -- the solved side of an interface sends 3 rows of 2048 nodes (32 KiB each), the slaved side 4; with pairs
    # (TM_TRIPLES_MIN_NODES=-1, or smaller blocks) a depth-2 halo per pair: 2 and 3 rows; end ranks have one neighbour
    mesh = _strip8()
    for triples in (True, False):
        if not triples:
            monkeypatch.setenv("TM_TRIPLES_MIN_NODES", "-1")
        for r in range(8):
            t = _table(mesh, list(range(8)), r, 8)
            assert t["direct"] and t["peer"] == [p for p in (r - 1, r + 1) if 0 <= p < 8]
            for k, b in enumerate(t["peer"]):
                assert t["send_cnt"][k] == ((4 if b < r else 3) if triples else (3 if b < r else 2)) * 2048
                assert t["recv_cnt"][k] == ((3 if b < r else 4) if triples else (2 if b < r else 3)) * 2048


def test_bad_arguments_are_refused():
    mesh = _strip8(64)
    md = _capi.MeshDesc(mesh, with_coordinates=False)
    t = _capi.tm_rccl_peer_table()
    own = (C.c_int32 * 8)(*([0] * 7 + [5]))
    assert _capi.lib().tm_rccl_peer_table_build(md.ref(), own, 0, 2, C.byref(t)) == _capi.TM_E_ARG
    assert _capi.lib().tm_rccl_peer_table_build(md.ref(), own, 2, 2, C.byref(t)) == _capi.TM_E_ARG


def test_triple_halo_message_sizes(monkeypatch):
    # 4096^2 blocks: sweep TRIPLES across ranks, a depth-3 halo, one exchange per triple -- the solved side of an interface sends 3
    # rows, the slaved side 4 (one more than for pairs on either side)
    mesh = configs.strip(8, 4096, 4096, only_blocks=set())
    for r in (0, 3, 7):
        t = _table(mesh, list(range(8)), r, 8)
        assert t["direct"]
        for k, b in enumerate(t["peer"]):
            assert t["send_cnt"][k] == (4 if b < r else 3) * 4096 and t["recv_cnt"][k] == (3 if b < r else 4) * 4096
    monkeypatch.setenv("TM_TRIPLES_MIN_NODES", "-1")   # off: the pairs' depth-2 halo
    t = _table(mesh, list(range(8)), 3, 8)
    assert [c // 4096 for c in t["send_cnt"]] == [3, 2]
    monkeypatch.setenv("TM_TRIPLES_MIN_NODES", "1")    # tests: any block of at least 16 x 16
    small = configs.strip(3, 40, 300, only_blocks=set())
    t = _table(small, [0, 1, 2], 1, 3)
    assert [c // 300 for c in t["send_cnt"]] == [4, 3]
