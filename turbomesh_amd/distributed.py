"""Multi-GPU driver: one process per GPU, one (group of) block(s) per rank.

The device solver loop lives in libtm_hip.so; a rank becomes part of a multi-GPU job through
`tm_comm_hooks` (include/tm_hip.h): the library packs the interface / first-interior rows its peers
need, calls `exchange`, and sums its reduction scalars with `allreduce_sum`.  This module implements
the two hooks with torch.distributed -- backend "nccl" is RCCL on ROCm, point-to-point over xGMI --
on views of ONE torch workspace tensor that backs every device buffer of the handle.

Exchange per operator application: one grouped isend/irecv pair per neighbouring rank
(rows are double2 = 16 B: 64 KiB per direction for a 4096-node interface); all-reduce: 4 doubles.
Both are latency-bound, so they are enqueued asynchronously on the stream the kernels run on and
never synchronise the host.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi, configs
from .smoothing import smooth, solver as _solver


def strip_for_rank(world, rank, ni, nj, blocks_per_rank=1):
    """Strip workload (configs.strip) with coordinates only for the blocks this rank owns."""
    nb = world * blocks_per_rank
    owned = set(range(rank * blocks_per_rank, (rank + 1) * blocks_per_rank))
    return configs.strip(nb, ni, nj, only_blocks=owned)


def local_plan(mesh, owner, rank, world):
    """Host-only (no GPU): the rank-local plan of tm_plan_local as numpy arrays."""
    md = _capi.MeshDesc(mesh, with_coordinates=False)
    info = _capi.tm_plan_local_info()
    own = (C.c_int32 * len(owner))(*owner)
    _capi.check(_capi.lib().tm_plan_local(md.ref(), own, rank, world, C.byref(info)))
    try:
        def arr(ptr, n):
            return np.ctypeslib.as_array(ptr, (n,)).copy() if n else np.zeros(0, dtype=np.int64)

        return {
            "n_owned": int(info.n_owned), "n_ghost": int(info.n_ghost), "n_send": int(info.n_send),
            "owned_blocks": arr(info.owned_blocks, info.nowned_blocks), "local_start": arr(info.local_start, info.nowned_blocks),
            "ghost_gid": arr(info.ghost_gid, info.n_ghost), "send_ids": arr(info.send_ids, info.n_send), "send_gid": arr(info.send_gid, info.n_send),
            "peer_rank": arr(info.peer_rank, info.npeers), "send_offset": arr(info.send_offset, info.npeers),
            "send_count": arr(info.send_count, info.npeers), "recv_offset": arr(info.recv_offset, info.npeers),
            "recv_count": arr(info.recv_count, info.npeers), "send_first": arr(info.send_first, info.npeers),
            "direct_send": bool(info.direct_send),
            "ghost_row_gid": arr(info.ghost_row_gid, info.n_ghost_rows),
            "ghost_row_kind": np.ctypeslib.as_array(info.ghost_row_kind, (info.n_ghost_rows,)).copy() if info.n_ghost_rows else np.zeros(0, dtype=np.int32),
            "ghost_row_cols": (np.ctypeslib.as_array(info.ghost_row_cols, (info.n_ghost_rows * 9,)).reshape(-1, 9).copy() if info.n_ghost_rows
                               else np.zeros((0, 9), dtype=np.int64)),
        }
    finally:
        _capi.lib().tm_plan_local_free(C.byref(info))


class HaloExchanger:
    """The communication pattern of one rank, independent of where the buffers live (CUDA or CPU tensors):
    start(send, recv) posts one grouped isend/irecv pair per neighbouring rank (peer k's slice of `send` goes to that peer,
    its rows land in peer k's slice of `recv`), wait() completes them; allreduce(t) sums in place.
    Rows are double2, so offsets/counts are scaled by 2 doubles.  The op lists are cached per (send, recv) buffer pair:
    the library alternates between a few fixed buffers, and building P2POps costs as much host time as a small kernel."""

    def __init__(self, plan, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.peers = [int(p) for p in plan["peer_rank"]]
        self.send = [(2 * int(o), 2 * int(c)) for o, c in zip(plan["send_offset"], plan["send_count"])]
        self.recv = [(2 * int(o), 2 * int(c)) for o, c in zip(plan["recv_offset"], plan["recv_count"])]
        self._ops = {}
        self._works = []

    def _build(self, send, recv):
        dist = self.dist
        ops = []
        for peer, (so, sc), (ro, rc) in zip(self.peers, self.send, self.recv):
            if rc:
                ops.append(dist.P2POp(dist.irecv, recv[ro:ro + rc], peer, self.group))
            if sc:
                ops.append(dist.P2POp(dist.isend, send[so:so + sc], peer, self.group))
        return ops

    def _staged(self, t):
        """gloo has no CUDA point-to-point: device buffers are staged through the host (debug / rehearsal transport)."""
        return t is not None and t.is_cuda and self.dist.get_backend(self.group) != "nccl"

    def start(self, send, recv):
        if self._staged(send) or self._staged(recv):
            self._recv_dev = recv
            send = send.cpu() if send is not None else None   # synchronises the stream: the pack kernel has finished
            recv = self._recv_host = (recv.cpu() if recv is not None else None)
            self._works = [self.dist.batch_isend_irecv(self._build(send, recv))] if (send is not None or recv is not None) else []
            self._works = [w for ws in self._works for w in ws]
            return
        self._recv_dev = None
        key = (send.data_ptr() if send is not None else 0, recv.data_ptr() if recv is not None else 0)
        ops = self._ops.get(key)
        if ops is None:
            ops = self._ops[key] = self._build(send, recv)
        self._works = self.dist.batch_isend_irecv(ops) if ops else []

    def wait(self):
        for w in self._works:
            w.wait()   # NCCL: makes the current stream wait, the host does not block
        self._works = []
        if getattr(self, "_recv_dev", None) is not None:
            self._recv_dev.copy_(self._recv_host)
            self._recv_dev = None

    def exchange(self, send, recv):
        self.start(send, recv)
        self.wait()

    def allreduce(self, t):
        if self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


class HooksBase:
    """ctypes plumbing of tm_comm_hooks on top of a torch workspace tensor."""

    def __init__(self, mesh, owner, rank, world, option=None, control=None, device=None):
        import torch

        self.torch = torch
        self.rank, self.world = rank, world
        self._owner = (C.c_int32 * len(owner))(*owner)
        self._exc = None
        self._exchange_cb = _capi.EXCHANGE_FN(self._on_exchange)
        self._allreduce_cb = _capi.ALLREDUCE_FN(self._on_allreduce)
        split = type(self).exchange_wait is not HooksBase.exchange_wait   # subclass implements the split form
        self._wait_cb = _capi.EXCHANGE_WAIT_FN(self._on_exchange_wait) if split else _capi.EXCHANGE_WAIT_FN()
        self._views = {}
        self._stream_ctx = {}
        hooks = _capi.tm_comm_hooks(None, rank, world, self._owner, self._exchange_cb, self._allreduce_cb, self._wait_cb, None, 0)
        opt = (option or _solver.Option.hip()).c_struct()
        from .smoothing import wall_control_function as _wcf

        cf = (control or _wcf.Algorithm.laplace()).c_struct()
        md = _capi.MeshDesc(mesh)
        nbytes = C.c_uint64(0)
        _capi.check(_capi.lib().tm_smoother_workspace_bytes(md.ref(), C.byref(opt), C.byref(cf), C.byref(hooks), C.byref(nbytes)))
        device = device or torch.device("cuda", torch.cuda.current_device())
        self.workspace = torch.zeros(int(nbytes.value) // 8 + 64, dtype=torch.float64, device=device)
        self._base = self.workspace.data_ptr()
        hooks.workspace = self._base
        hooks.workspace_bytes = self.workspace.numel() * 8
        self._hooks = hooks
        self.smoother = smooth.Smoother(mesh, option, control, hooks=hooks, stream=torch.cuda.current_stream(device).cuda_stream)
        self.plan = self._read_plan()
        # rows of the send buffer the hooks may touch: a packed buffer, or the vector itself (tm_smoother_exchange_plan)
        self.n_send = int(max((o + c for o, c in zip(self.plan["send_offset"], self.plan["send_count"])), default=0))
        self.n_ghost = int(sum(self.plan["recv_count"]))

    def _read_plan(self):
        n = C.c_int32(0)
        pr = C.POINTER(C.c_int32)()
        so, sc, ro, rc = (C.POINTER(C.c_int64)() for _ in range(4))
        _capi.check(_capi.lib().tm_smoother_exchange_plan(self.smoother._h, C.byref(n), C.byref(pr), C.byref(so), C.byref(sc), C.byref(ro), C.byref(rc)))
        k = n.value

        def arr(ptr):
            return np.ctypeslib.as_array(ptr, (k,)).copy() if k else np.zeros(0, dtype=np.int64)

        return {"peer_rank": arr(pr), "send_offset": arr(so), "send_count": arr(sc), "recv_offset": arr(ro), "recv_count": arr(rc)}

    def _view(self, ptr, ndoubles):
        key = (int(ptr), ndoubles)
        v = self._views.get(key)
        if v is None:
            off = (int(ptr) - self._base) // 8
            assert 0 <= off and off + ndoubles <= self.workspace.numel(), "hook pointer outside the workspace"
            v = self._views[key] = self.workspace[off:off + ndoubles]
        return v

    def _on_stream(self, stream):
        """The library names the stream every hook call belongs to (a sweep pair's exchanges run on the handle's side stream,
        beside the interior pass): make it torch's current stream, so that copies and NCCL work are ordered against it."""
        key = int(stream or 0)
        ctx = self._stream_ctx.get(key)
        if ctx is None:
            dev = self.workspace.device
            ctx = self._stream_ctx[key] = self.torch.cuda.ExternalStream(key, device=dev) if key else self.torch.cuda.default_stream(dev)
        return self.torch.cuda.stream(ctx)

    def _on_exchange(self, ctx, send_ptr, recv_ptr, stream):
        try:
            with self._on_stream(stream):
                self.exchange(self._view(send_ptr, 2 * self.n_send) if self.n_send else None, self._view(recv_ptr, 2 * self.n_ghost) if self.n_ghost else None)
            return 0
        except BaseException as e:   # never let an exception cross the C boundary
            self._exc = e
            return 1

    def _on_exchange_wait(self, ctx, stream):
        try:
            with self._on_stream(stream):
                self.exchange_wait()
            return 0
        except BaseException as e:
            self._exc = e
            return 1

    def _on_allreduce(self, ctx, buf, n, stream):
        try:
            with self._on_stream(stream):
                self.allreduce(self._view(buf, int(n)))
            return 0
        except BaseException as e:
            self._exc = e
            return 1

    def iterate(self, iterations):
        try:
            return self.smoother.iterate(iterations)
        except _capi.TmError:
            if self._exc is not None:
                raise self._exc
            raise

    # subclasses implement the transport; overriding exchange_wait selects the split (overlapped) form
    def exchange(self, send, recv):
        raise NotImplementedError

    def exchange_wait(self):
        raise NotImplementedError

    def allreduce(self, t):
        raise NotImplementedError


class TorchHooks(HooksBase):
    """tm_comm_hooks over torch.distributed (RCCL p2p + all-reduce)."""

    def __init__(self, mesh, owner, rank, world, option=None, control=None, group=None, device=None):
        super().__init__(mesh, owner, rank, world, option, control, device)
        self._x = HaloExchanger(self.plan, group)

    def exchange(self, send, recv):      # starts the transfer; K2 runs while it is in flight
        self._x.start(send, recv)

    def exchange_wait(self):
        self._x.wait()

    def allreduce(self, t):
        self._x.allreduce(t)


class RcclHooks:
    """tm_comm_hooks served by the library's own RCCL transport (tm_rccl_*): torch.distributed is used once, to hand
    rank 0's ncclUniqueId to the other ranks; after that no Python runs inside a sweep (a batch_isend_irecv per exchange
    costs 50-70 us of host time, tools/ubench/p2p_host_cost.py -- more than half of a two-sweep pass of a 4096^2 block)."""

    @staticmethod
    def _librccl_path():
        """The RCCL this process should use: $TM_RCCL_LIB when set (a site's own build), else the copy torch ships and has already
        loaded (one RCCL per process), else the library's default search (NULL)."""
        import os

        import torch

        env = os.environ.get("TM_RCCL_LIB")
        if env:
            return env.encode()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return path.encode() if os.path.exists(path) else None

    @staticmethod
    def precheck():
        """Everything a rank can check ALONE before the first collective call: librccl loads, its symbols resolve and a
        unique id can be made.  Ranks vote on this (all_reduce MIN) before any of them enters ncclCommInitRank."""
        uid = (C.c_ubyte * 128)()
        _capi.check(_capi.lib().tm_rccl_unique_id(RcclHooks._librccl_path(), uid))

    def __init__(self, mesh, owner, rank, world, option=None, control=None, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.rank, self.world = rank, world
        L = _capi.lib()
        self._path = self._librccl_path()
        device = device or torch.device("cuda", torch.cuda.current_device())
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            _capi.check(L.tm_rccl_unique_id(self._path, uid))
        if world > 1:
            on_gpu = dist.get_backend(group) == "nccl"
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=device if on_gpu else "cpu")
            dist.broadcast(t, src=0, group=group)
            uid = (C.c_ubyte * 128)(*t.cpu().tolist())
        self._comm = C.c_void_p()
        _capi.check(L.tm_rccl_comm_create(self._path, uid, rank, world, C.byref(self._comm)))
        self._owner = (C.c_int32 * len(owner))(*owner)
        self._md = _capi.MeshDesc(mesh)
        self._hooks = _capi.tm_comm_hooks()
        # tables for THIS handle's options: a handle that never runs sweep triples exchanges the depth-2 halo only
        from .smoothing import wall_control_function as _wcf

        self._opt_c = (option or _solver.Option.hip()).c_struct()
        self._cf_c = (control or _wcf.Algorithm.laplace()).c_struct()
        _capi.check(L.tm_rccl_hooks_for(self._comm, self._md.ref(), self._owner, C.byref(self._opt_c), C.byref(self._cf_c), C.byref(self._hooks)))
        self.smoother = smooth.Smoother(mesh, option, control, hooks=self._hooks, stream=torch.cuda.current_stream(device).cuda_stream)

    def iterate(self, iterations):
        return self.smoother.iterate(iterations)

    def close(self):
        if getattr(self, "smoother", None) is not None:
            self.smoother.close()
            self.smoother = None
        if getattr(self, "_comm", None):
            _capi.lib().tm_rccl_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
