"""Shared small topologies for the parity tests (all built with turbomesh_amd.configs)."""
from turbomesh_amd import configs

# name -> builder(tfi) ; sizes chosen so the oracle + scipy finish in well under a second
TOPOLOGIES = {
    "single_17x21": lambda tfi=None: configs.single_block(17, 21, tfi=tfi),
    "single_perturbed_33": lambda tfi=None: configs.single_block(33, 33, tfi=tfi, perturb=0.25),
    "single_ragged_70x131": lambda tfi=None: configs.single_block(70, 131, tfi=tfi),
    "strip3_9x12": lambda tfi=None: configs.strip(3, 9, 12, tfi=tfi),
    "strip3_reversed": lambda tfi=None: configs.strip(3, 9, 12, tfi=tfi, reverse_odd=True),
    "strip2_40x300": lambda tfi=None: configs.strip(2, 40, 300, tfi=tfi),
    "channel_periodic_sliding": lambda tfi=None: configs.periodic_channel(13, 9, tfi=tfi),
    "channel_periodic_fixed": lambda tfi=None: configs.periodic_channel(21, 15, tfi=tfi, sliding=False),
    "two_by_two_junction": lambda tfi=None: configs.two_by_two(8, 9, tfi=tfi),
    "plate_le": lambda tfi=None: configs.plate(15, 9, tfi=tfi),
}


def cut(grid, isplits=(), jsplits=(), flip_i=(), flip_j=(), swap=()):
    """Cut an (NI, NJ, 2) coordinate array into (len(isplits)+1) x (len(jsplits)+1) blocks that share their interface nodes, coupled by one
    connection per internal interface -- the SAME grid as one block or as several, for partition-invariance tests.
    Block (a, b) is number a * nbj + b.  flip_i / flip_j: block numbers stored with that index direction reversed (their ranges on the
    interfaces then run backwards and the side names swap); swap: interface numbers (in creation order) whose two ranges trade places
    (which side is solved and which is slaved, smooth.zig:1029-1084).
    Side names as in turbomesh_amd.configs: j_min / j_max = the rows i = 0 / ni-1, i_min / i_max = the columns j = 0 / nj-1."""
    import numpy as np

    from turbomesh_amd import configs
    from turbomesh_amd.boundary import Connection, Range, Side
    from turbomesh_amd.discrete import Mesh

    NI, NJ = grid.shape[:2]
    ib = [0] + list(isplits) + [NI - 1]
    jb = [0] + list(jsplits) + [NJ - 1]
    nbi, nbj = len(ib) - 1, len(jb) - 1
    m = Mesh()
    shape = {}
    for a in range(nbi):
        for b in range(nbj):
            k = a * nbj + b
            sub = grid[ib[a]:ib[a + 1] + 1, jb[b]:jb[b + 1] + 1]
            if k in flip_i:
                sub = sub[::-1]
            if k in flip_j:
                sub = sub[:, ::-1]
            sub = np.ascontiguousarray(sub).copy()
            shape[k] = sub.shape[:2]
            m.addBlock(f"cut_{a}_{b}", configs.block_from_array(sub))

    def rng(k, plus_side, along_i):
        """Range of block k on its physical +side (plus_side) or -side, enumerating the physical index ascending."""
        ni, nj = shape[k]
        if along_i:   # a column of the block (j = const): the interface between (a, b) and (a, b + 1)
            at_max = plus_side != (k in flip_j)
            side = Side.i_max if at_max else Side.i_min
            n, rev = ni, k in flip_i
        else:         # a row of the block (i = const): the interface between (a, b) and (a + 1, b)
            at_max = plus_side != (k in flip_i)
            side = Side.j_max if at_max else Side.j_min
            n, rev = nj, k in flip_j
        return Range(k, side, n - 1 if rev else 0, 0 if rev else n - 1)

    nconn = 0
    for a in range(nbi):
        for b in range(nbj):
            k = a * nbj + b
            pairs = []
            if a + 1 < nbi:
                pairs.append((rng(k, True, False), rng((a + 1) * nbj + b, False, False)))
            if b + 1 < nbj:
                pairs.append((rng(k, True, True), rng(a * nbj + b + 1, False, True)))
            for r0, r1 in pairs:
                m.connections.append(Connection((r1, r0) if nconn in swap else (r0, r1), None))
                nconn += 1
    return m


def uncut(mesh, NI, NJ, isplits=(), jsplits=(), flip_i=(), flip_j=()):
    """Inverse of cut(): the blocks' coordinates put back into one (NI, NJ, 2) array (interface nodes taken from the later block)."""
    import numpy as np

    ib = [0] + list(isplits) + [NI - 1]
    jb = [0] + list(jsplits) + [NJ - 1]
    nbj = len(jb) - 1
    out = np.full((NI, NJ, 2), np.nan)
    for a in range(len(ib) - 1):
        for b in range(nbj):
            k = a * nbj + b
            blk = mesh.blocks[k]
            sub = blk.points.data if hasattr(blk, "points") else blk
            if k in flip_j:
                sub = sub[:, ::-1]
            if k in flip_i:
                sub = sub[::-1]
            out[ib[a]:ib[a + 1] + 1, jb[b]:jb[b + 1] + 1] = sub
    return out
