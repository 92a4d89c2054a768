"""The library's own RCCL transport (tm_rccl_*) on the one GPU this box has: a 1-rank communicator created from a
unique id, hooks filled by the library, and a handle driven through them must reproduce the hook-less handle bit for bit.
(The N > 1 exchange pattern itself is covered by the virtual-rank tests with the same plan tables, and by bench.py's
start-up cross-check against the torch.distributed transport on the 8-GPU run.)"""
import ctypes as C

import numpy as np
import pytest

from tests.conftest import mesh_flat
from turbomesh_amd import _capi, configs, distributed as tmd
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu


def test_rccl_hooks_single_rank_equal_plain_handle():
    build = lambda: configs.strip(2, 40, 130, reverse_odd=True)
    for opt, its in ((solver.Option.hip(inner=solver.Inner.relax), 7), (solver.Option.hip(rtol=1e-13, eager_scalars=True), 2)):   # eager: the recurrence a hooked handle runs
        ref = build()
        smooth.mesh(ref, its, opt)
        got = build()
        h = tmd.RcclHooks(got, owner=[0, 0], rank=0, world=1, option=opt)
        st = h.iterate(its)
        h.smoother.download()
        h.close()
        assert st["outer_iterations"] == its
        assert np.array_equal(mesh_flat(got), mesh_flat(ref))


def test_rccl_argument_checks():
    L = _capi.lib()
    comm = C.c_void_p()
    uid = (C.c_ubyte * 128)()
    assert L.tm_rccl_comm_create(None, uid, 3, 2, C.byref(comm)) == _capi.TM_E_ARG          # rank outside the job
    assert L.tm_rccl_unique_id(None, None) == _capi.TM_E_ARG
    hooks = _capi.tm_comm_hooks()
    assert L.tm_rccl_hooks(None, None, None, C.byref(hooks)) == _capi.TM_E_ARG
