"""CPU: libtm_hip.so loads and exports every function include/tm_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from turbomesh_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(tm_[a-z0-9_]+)\s*\(", src))
    names -= {"tm_comm_hooks"}
    return sorted(names)


DIAGNOSTIC = ["tm_csr_ilu0_probe", "tm_smoother_profile", "tm_smoother_profile_read", "tm_smoother_queue_ordering", "tm_stream_probe", "tm_white_math_probe"]


def _declared_functions():
    # the drop-in surface (tm_hip.h) + the measurement / diagnostic entry points (tm_hip_diag.h), both exported by the one product library
    return sorted(set(_declared_in("tm_hip.h")) | set(_declared_in("tm_hip_diag.h")))


def test_diagnostics_live_in_their_own_header():
    # nothing a binding of the reference needs is a measurement helper, and no measurement helper sits in the drop-in header
    assert _declared_in("tm_hip_diag.h") == DIAGNOSTIC
    assert not set(DIAGNOSTIC) & set(_declared_in("tm_hip.h"))


def test_header_symbols_exported():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 20
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(_capi.EXPORTS)) == declared, set(declared) ^ set(_capi.EXPORTS)


def test_product_library_exports_nothing_the_header_does_not_declare():
    # the measurement helpers (tm_debug_*, tm_tune_*, tm_diag_*) live in libtm_hip_dbg.so only (csrc/Makefile, -DTM_DEBUG_EXPORTS)
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("tm_")})
    assert exported == _declared_functions(), set(exported) ^ set(_declared_functions())
    if os.path.exists(_capi.DBG_LIB_PATH):
        dbg = ctypes.CDLL(_capi.DBG_LIB_PATH)
        assert all(hasattr(dbg, n) for n in _declared_functions() + ["tm_debug_null_hooks", "tm_tune_apply", "tm_diag_apply"])


def test_abi_version_and_error_string_without_gpu():
    lib = _capi.lib()
    assert lib.tm_abi_version() == 1
    assert isinstance(lib.tm_last_error(), bytes)


def test_struct_layouts_match_header():
    # sizes of the POD mirrors (x86-64): tm_range 32, tm_connection 88, tm_condition 40, tm_block 24, tm_mesh_desc 48
    assert ctypes.sizeof(_capi.tm_range) == 32 and ctypes.sizeof(_capi.tm_connection) == 88
    assert ctypes.sizeof(_capi.tm_condition) == 40 and ctypes.sizeof(_capi.tm_block) == 24 and ctypes.sizeof(_capi.tm_mesh_desc) == 48
    assert ctypes.sizeof(_capi.tm_solver_opt) == 48 and ctypes.sizeof(_capi.tm_stats) == 72 and ctypes.sizeof(_capi.tm_control_fn) == 24


def test_product_has_no_oracle_dependency():
    # the product path must never import, link, include or dlopen anything under oracle/
    pkg = os.path.join(ROOT, "turbomesh_amd")
    forbidden = re.compile(r"(import\s+oracle|from\s+oracle|tm_oracle\.h|libtm_oracle|\borc_[a-z_]+\(|oracle/)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                txt = re.sub(r"//[^\n]*", "", txt) if not f.endswith(".py") else re.sub(r"(?m)^\s*#[^\n]*", "", txt)   # comments may cite the oracle
                assert not forbidden.search(txt), os.path.join(dirpath, f)


@pytest.mark.gpu
def test_gfx950_required_message():
    # on the GPU box the library accepts the device; this only checks the happy path does not raise
    import numpy as np

    from turbomesh_amd import configs

    m = configs.single_block(5, 5)
    assert not np.isnan(m.blocks[0].points.data).any()
