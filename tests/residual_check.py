"""Kernel-independent check of a Picard iterate: the TRUE residual of the GPU's X^1 against the system the faithful oracle assembles from X^0.

The oracle (oracle/orc_system.cpp: RowCompressedMatrixSystem2d.init + fill, smooth.zig:309-385, 923-1113) builds the reference's CSR -- nine
coefficients per interior row in the reference's expression order, the static perimeter rows, the right-hand sides -- without any of the
device's code (no plan table, no factored stencil, no matrix-free kernel).  With the device's iterate X^1 the scaled residual

    || D^-1 (b - A(X^0) X^1) ||_2 / || D^-1 b ||_2 ,      D = diag A(X^0)

is formed on the host in 80-bit extended precision (x86 long double: the products of fp64 numbers are exact in it and a row's nine terms
accumulate with 2^-64 relative error), because in fp64 the EVALUATION of the residual has a floor of ~ eps sqrt(rows) |x| / ||D^-1 b|| ~ 1e-14
at 4096^2 -- above the tolerance the library stops its recurrence residual at (7.5e-9 / nodes = 4.5e-16 there).  Test infrastructure."""
import numpy as np


class _Single:
    """Duck-typed oracle mesh: blocks with fixed walls, no connections (the oracle accepts that; the reference underflows)."""

    def __init__(self, blocks):
        self.blocks = [np.array(b, dtype=np.float64, order="C", copy=True) for b in blocks]
        self.connections = []
        self.bcs = []


def assemble(x0_blocks):
    """The reference's system at X^0 as (indptr int32, indices int32, values f64, rhs (n, 2) f64), assembled by the faithful oracle."""
    from oracle import oracle

    s = oracle.System(_Single(x0_blocks))
    s.fill(0)
    p, i, v = s.lhs_p.copy(), s.lhs_i.copy(), s.lhs_values.copy()
    b = np.stack([s.rhs_x, s.rhs_y], axis=1).copy()
    s.close()
    return p, i, v, b


def scaled_residual(p, i, v, b, x, dtype=np.longdouble, chunk_rows=1 << 20, floor=None):
    """(|| D^-1 (b - A x) ||_2, || D^-1 b ||_2) per component, evaluated in `dtype`; x, b are (n, 2).  Rows are visited in chunks so that the
    temporaries stay at ~ chunk_rows x 9 extended-precision numbers.  floor (a 2-vector, optional) receives || D^-1 |A| |x| ||_2 per
    component: times the unit roundoff it is the residual a vector picks up from merely being STORED in fp64 (each x_k off by up to
    eps |x_k|), i.e. what no fp64 solver can get under."""
    n = len(p) - 1
    num = np.zeros(2, dtype=dtype)
    den = np.zeros(2, dtype=dtype)
    flo = np.zeros(2, dtype=dtype)
    xw = x.astype(dtype)
    for r0 in range(0, n, chunk_rows):
        r1 = min(n, r0 + chunk_rows)
        lo, hi = int(p[r0]), int(p[r1])
        cols = i[lo:hi]
        vals = v[lo:hi].astype(dtype)
        starts = (p[r0:r1] - lo).astype(np.int64)
        counts = np.diff(np.append(starts, hi - lo))
        rows = np.repeat(np.arange(r0, r1, dtype=np.int64), counts)
        diag = np.zeros(r1 - r0, dtype=dtype)
        on = cols == rows
        diag[rows[on] - r0] = vals[on]
        assert (diag != 0).all(), "a row without a diagonal entry"
        for c in range(2):
            ax = np.add.reduceat(vals * xw[cols, c], starts)
            res = (b[r0:r1, c].astype(dtype) - ax) / diag
            num[c] += np.dot(res, res)
            sb = b[r0:r1, c].astype(dtype) / diag
            den[c] += np.dot(sb, sb)
            if floor is not None:
                aa = np.add.reduceat(np.abs(vals) * np.abs(xw[cols, c]), starts) / np.abs(diag)
                flo[c] += np.dot(aa, aa)
    if floor is not None:
        floor[:] = np.sqrt(flo).astype(np.float64)
    return np.sqrt(num), np.sqrt(den)


def relative_residual(p, i, v, b, x1, dtype=np.longdouble, with_floor=False):
    """The library's own stop quantity, components together: sqrt(sum_c ||D^-1 r_c||^2) / sqrt(sum_c ||D^-1 b_c||^2), and per component.
    with_floor: a third value, the fp64 storage floor of that quantity: 2^-53 sqrt(sum_c ||D^-1 |A| |x_c|||^2) / the same denominator."""
    flo = np.zeros(2) if with_floor else None
    num, den = scaled_residual(p, i, v, b, x1, dtype, floor=flo)
    d = np.sqrt((den ** 2).sum())
    both = float(np.sqrt((num ** 2).sum()) / d)
    per = [float(num[c] / den[c]) for c in range(2)]
    if with_floor:
        return both, per, float(2.0 ** -53 * np.sqrt((flo ** 2).sum()) / float(d))
    return both, per
