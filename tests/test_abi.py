"""C-ABI checks that need no GPU: the library loads, exports every symbol include/nmfx.h declares, the
ctypes structs match the header layout, and the product fails loudly (never falls back) without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "nmfx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nmfx_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = _header_symbols()
    for s in ("nmfx_create", "nmfx_set_X", "nmfx_solve", "nmfx_iterate", "nmfx_destroy", "nmfx_last_error",
              "nmfx_comm_init", "nmfx_comm_get_unique_id", "nmfx_alspgrad_subsolve", "nmfx_objective"):
        assert s in syms


def test_library_exports_every_declared_symbol(built):
    import nmfx
    lib = nmfx._lib.load()
    declared = _header_symbols()
    assert sorted(nmfx._lib.SYMBOLS) == declared
    for s in declared:
        assert getattr(lib, s) is not None
    assert b"gfx950" in lib.nmfx_version()


def test_struct_layouts(built):
    import nmfx
    L = nmfx._lib
    assert C.sizeof(L.Opts) == 6 * 4 + 11 * 8 + 6 * 4
    assert C.sizeof(L.CResult) == 8 + 4 + 4 + 8 + 8 + 8 + 8 + 8
    assert C.sizeof(L.KernelStat) == 64 + 8 + 8 + 8 + 8
    assert L.Opts.tol.offset == 24 and L.CResult.objvalue.offset == 16


def test_no_cpu_fallback(built):
    """Without a GPU nmfx_create must return NO_DEVICE with a message -- the product never routes to the oracle."""
    import torch
    import nmfx
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nmfx.NMFXError) as e:
        nmfx.Context("float32", 8, 8, 2)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    """The shipped package (nmf.jl_amd/) must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, "nmf.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".jl")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "nmf_oracle" not in txt and "c_oracle" not in txt, os.path.join(dp, f)


def test_every_environment_switch_of_the_library_is_documented():
    """INTEGRATION.md lists every NMFX_* variable the library reads -- the development switches behind NMFX_DEV=1 (csrc/comm.hpp: dev_env)
    and the user switches read with getenv -- so that a maintainer can tell from the document alone which code paths exist beside the
    defaults."""
    csrc = os.path.join(ROOT, "nmf.jl_amd", "csrc")
    names = set()
    for f in os.listdir(csrc):
        if f.endswith((".hpp", ".hip")):
            txt = open(os.path.join(csrc, f), errors="ignore").read()
            names |= set(re.findall(r'(?:dev_env|getenv)\("(NMFX_[A-Z0-9_]+)"\)', txt))
    assert len(names) >= 30      # (the pattern still matches the source)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
