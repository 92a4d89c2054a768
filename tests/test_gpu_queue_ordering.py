"""The two queues of a pipelined pass (interior pass on the handle's stream, perimeter-row chain on a high-priority stream of the handle's own)
are ordered by counters in device memory + one-wave announce / wait kernels -- which only works when the two streams sit on different
hardware queues.  HIP does not promise that, so the handle tests it when the second stream is created and falls back to events
(Smoother::queue_self_test, include/tm_hip_diag.h: tm_smoother_queue_ordering).  Here: the decision is reported, it is the right one when
both streams are FORCED onto one hardware queue, and the coordinates are bit-identical to single sweeps either way."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import mesh_flat
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
import torch
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
high = sys.argv[1] == "high"
build = lambda: configs.strip(2, 768, 768)       # 1.18 M nodes in one process: coupled sweep triples on two queues
ref = build()
smooth.mesh(ref, 10, solver.Option.hip(inner=solver.Inner.relax, single_sweep=True))
got = build()
st = torch.cuda.Stream(priority=-1) if high else torch.cuda.Stream()
with smooth.Smoother(got, solver.Option.hip(inner=solver.Inner.relax), stream=st.cuda_stream) as sm:
    sm.iterate(10)
    code, what = sm.queue_ordering()
    sm.download()
same = all(np.array_equal(a.points.data, b.points.data) for a, b in zip(got.blocks, ref.blocks))
print(json.dumps({"code": code, "what": what, "bit_identical": bool(same)}))
"""


def _run(arg, **env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("TM_PAIR_SYNC", "GPU_MAX_HW_QUEUES"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT, arg], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_counters_when_the_streams_run_side_by_side():
    got = configs.strip(2, 768, 768)
    with smooth.Smoother(got, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        assert sm.queue_ordering()[0] == -1          # no two-queue pass has run yet
        sm.iterate(9)
        assert sm.queue_ordering() == (0, "counters")


def test_a_handle_with_fixed_walls_has_no_second_queue():
    m = configs.single_block(300, 300)
    with smooth.Smoother(m, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        sm.iterate(6)
        assert sm.queue_ordering()[0] == -1


def test_events_on_request():
    j = _run("normal", TM_PAIR_SYNC="events")
    assert j["code"] == 1 and j["bit_identical"]


def test_self_test_selects_events_when_both_streams_share_one_hardware_queue():
    # GPU_MAX_HW_QUEUES=1 leaves the runtime ONE hardware queue per priority class; the caller's stream is a high-priority one, like the
    # handle's chain stream, so both land on it.  The first announce-and-wait kernel then runs into its 5 ms limit (the second cannot
    # start behind it), the handle reports 3 and orders every pass with events: same bits as single sweeps.
    j = _run("high", GPU_MAX_HW_QUEUES="1")
    assert j["code"] == 3, j
    assert j["bit_identical"]
    # the same caller's stream with the usual number of hardware queues: side by side, counters
    j2 = _run("high")
    assert j2["code"] == 0 and j2["bit_identical"], j2
