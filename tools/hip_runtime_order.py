#!/usr/bin/env python3
"""libtm_hip first, torch second: both must see the GPU (one HIP runtime per process, turbomesh_amd/_capi.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbomesh_amd import configs
from turbomesh_amd.smoothing import smooth, solver
mesh = configs.single_block(33, 41)
st = smooth.mesh(mesh, 2, solver.Option.hip())
import torch
print("library first: ok;", "torch sees", torch.cuda.device_count(), "device(s);", torch.zeros(3, device="cuda").sum().item())
