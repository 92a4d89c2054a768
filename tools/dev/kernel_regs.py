#!/usr/bin/env python3
"""Registers, spills and scratch per kernel of a built object (no GPU needed): extracts the gfx950 code object from the .o's .hip_fatbin
section and reads the kernel descriptors' metadata.   usage: kernel_regs.py [object = turbomesh_amd/csrc/.obj/tm_kernels.o] [name filter]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "turbomesh_amd", "csrc", ".obj", "tm_kernels.o")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for k in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", k)
    if not name:
        continue
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if flt and flt not in dem:
        continue
    g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, k).group(1))
    print(f"{dem[:70]:70s} vgpr {g('vgpr_count'):4d}  vgpr_spill {g('vgpr_spill_count'):4d}  sgpr {g('sgpr_count'):4d}  scratch {g('private_segment_fixed_size'):5d} B  lds {g('group_segment_fixed_size'):6d}")
