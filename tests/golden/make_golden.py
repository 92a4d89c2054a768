#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/*.npz in THIS container (no GPU, no reference build):

  inputs : edge points + clusterings (TFI cases); seeded block coordinates + topology (smoothing cases)
  outputs: the oracle's TFI field; the exact Picard iterates 1..3 and the reference-style residual history,
           computed as oracle-assembled CSR (reference smooth.zig:421-1165) + scipy.sparse.linalg.splu per component
           -- the semantics of the reference's UMFPACK backend (umfpack.zig:18-24).

The reference holds no golden vector for this path ("parity unpinned"); these fixtures pin the oracle AND the HIP path
against the cross-checked restatement.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests.conftest import OracleMesh, oracle_tfi  # noqa: E402
from turbomesh_amd import clustering, configs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def mesh_arrays(mesh):
    conns = np.array([[c.ranges[0].block, int(c.ranges[0].side), c.ranges[0].start, c.ranges[0].end, c.ranges[1].block, int(c.ranges[1].side),
                       c.ranges[1].start, c.ranges[1].end] for c in mesh.connections], dtype=np.int64).reshape(-1, 8)
    per = np.array([[np.nan, np.nan] if c.periodicity is None else list(c.periodicity) for c in mesh.connections], dtype=np.float64).reshape(-1, 2)
    bcs = np.array([[b.range.block, int(b.range.side), b.range.start, b.range.end, int(b.kind)] for b in mesh.boundary_conditions],
                   dtype=np.int64).reshape(-1, 5)
    return conns, per, bcs


def smoothing_case(name, mesh, control=None, iterations=3):
    om = OracleMesh(mesh)
    hist, iterates = oracle.picard_exact(om, iterations, control=control, keep_iterates=True)
    conns, per, bcs = mesh_arrays(mesh)
    out = {"conns": conns, "periodicity": per, "bcs": bcs, "residual_history": hist, "nblocks": np.int64(len(mesh.blocks)),
           "control": np.array([0.0, 0.0, 0.0] if control is None else [1.0, control[1], control[2]])}
    for b, blk in enumerate(mesh.blocks):
        out[f"seed_{b}"] = blk.points.data
        for k in range(iterations):
            out[f"iter{k + 1}_{b}"] = iterates[k][b]
    np.savez_compressed(os.path.join(HERE, f"smooth_{name}.npz"), **out)
    print(name, "dof", sum(b.points.data.shape[0] * b.points.data.shape[1] for b in mesh.blocks), "residuals", hist)


def tfi_case(name, ni, nj, cls):
    import math

    s1, s2, t1, t2 = (c.compute(n) for c, n in zip(cls, (ni, ni, nj, nj)))
    for c in (s1, s2, t1, t2):
        c[0], c[-1] = 0.0, 1.0
    i_min = np.stack([s1 * 2.0, 0.15 * np.sin(math.pi * s1)], axis=1)
    i_max = np.stack([s2 * 2.0 + 0.1 * np.sin(math.pi * s2), 1.0 + 0.2 * np.sin(2 * math.pi * s2)], axis=1)
    j_min = np.stack([-0.1 * np.sin(math.pi * t1), t1], axis=1)
    j_max = np.stack([2.0 + 0.1 * np.sin(math.pi * t2), t2], axis=1)
    j_min[0], j_min[-1] = i_min[0], i_max[0]
    j_max[0], j_max[-1] = i_min[-1], i_max[-1]
    field = oracle.tfi_block(i_min, i_max, j_min, j_max, s1, s2, t1, t2)
    np.savez_compressed(os.path.join(HERE, f"tfi_{name}.npz"), x_i_min=i_min, x_i_max=i_max, x_j_min=j_min, x_j_max=j_max, s1=s1, s2=s2, t1=t1, t2=t2,
                        field=field)
    print("tfi", name, field.shape)


if __name__ == "__main__":
    tfi_case("uniform_17x23", 17, 23, [clustering.Uniform()] * 4)
    tfi_case("mixed_33x41", 33, 41, [clustering.Uniform(), clustering.Roberts(0.5, 1.05), clustering.SingleHyperbolicClustering(0.002), clustering.Roberts(0.0, 1.2)])
    smoothing_case("single_17x21", configs.single_block(17, 21, tfi=oracle_tfi))
    smoothing_case("single_perturbed_33", configs.single_block(33, 33, tfi=oracle_tfi, perturb=0.25))
    smoothing_case("strip3_reversed", configs.strip(3, 9, 12, tfi=oracle_tfi, reverse_odd=True))
    smoothing_case("channel_periodic_sliding", configs.periodic_channel(13, 9, tfi=oracle_tfi))
    smoothing_case("two_by_two_junction", configs.two_by_two(8, 9, tfi=oracle_tfi))
    smoothing_case("plate_white", configs.plate(15, 9, tfi=oracle_tfi), control=("white", 0.02, 0.5 * np.pi), iterations=4)
