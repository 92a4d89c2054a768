"""BASELINE.json full sizes on one MI355X, checked through size-independent properties (the oracle would need
minutes/hours at these sizes): config 4 = 8 blocks of 2048^2 coupled by 7 interfaces."""
import numpy as np
import pytest

from oracle import oracle
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver

pytestmark = pytest.mark.gpu


def test_config4_eight_blocks_2048():
    nb, n = 8, 2048
    mesh = configs.strip(nb, n, n)   # TFI of every block on the GPU (K1), 8 x 64 MiB
    walls = [(b.points.data[:, 0].copy(), b.points.data[:, -1].copy()) for b in mesh.blocks]
    bottom, top = mesh.blocks[0].points.data[0].copy(), mesh.blocks[-1].points.data[-1].copy()
    win = [mesh.blocks[3].points.data[-40:, 500:633].copy(), mesh.blocks[4].points.data[:40, 500:633].copy()]   # window across interface 3|4
    with smooth.Smoother(mesh, solver.Option.hip(inner=solver.Inner.relax)) as sm:
        st1 = sm.iterate(1)
        sm.download()
        # fixed rows return the boundary coordinates bit-exactly (smooth.zig:790-795)
        for b, (w0, w1) in zip(mesh.blocks, walls):
            assert np.array_equal(b.points.data[:, 0], w0) and np.array_equal(b.points.data[:, -1], w1)
        assert np.array_equal(mesh.blocks[0].points.data[0], bottom) and np.array_equal(mesh.blocks[-1].points.data[-1], top)
        # interior nodes next to the interface: one oracle sweep of a window (device operation order) reproduces them bit for bit
        for w, blk, rows in ((win[0], mesh.blocks[3].points.data[-40:, 500:633], slice(1, 38)), (win[1], mesh.blocks[4].points.data[:40, 500:633], slice(2, 39))):
            ref = w.copy()
            oracle.time_relax_sweeps(ref, 1, 1.0)
            assert np.array_equal(blk[rows, 1:-1], ref[rows, 1:-1])
        st2 = sm.iterate(30)
        sm.download()
        assert 0 < st2["last_residual"] < st1["last_residual"]
        for k in range(nb - 1):
            a, b = mesh.blocks[k].points.data[-1], mesh.blocks[k + 1].points.data[0]
            assert not np.isnan(a).any()
            # slaved copy lags the solved copy by exactly one sweep: the gap is bounded by the last displacement
            assert np.abs(a - b).max() <= 4.0 * np.sqrt(st2["last_dx2"] + st2["last_dy2"]) + 1e-15
        assert st2["operator_sweeps"] == 30
