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
