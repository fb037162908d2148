"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, host-only
entry points work, and bad input is rejected with error codes (no compute, no GPU)."""
import ctypes
import re
import os

import numpy as np
import pytest

from pastix_amd import _lib, fact_flops
from conftest import golden_names, ROOT


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pastix_amd.h")).read()
    declared = set(re.findall(r"\b(pastix_amd_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    host = ""
    for h in ("pastix_amd_symbolic.h", "pastix_amd_driver.h"):
        host += open(os.path.join(ROOT, "include", h)).read()
    declared_host = set(re.findall(r"\b(pastix_amd_[a-z_0-9]+)\s*\(", host))
    assert declared_host == set(_lib.EXPORTS_HOST)
    declared |= declared_host
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.pastix_amd_version()


@pytest.mark.parametrize("name", golden_names())
def test_fact_flops_matches_reference(name, golden):
    g = golden(name)
    f = fact_flops(g["cblk4"], g["blok4"], g["facto"])
    assert abs(f - g["flops"]) <= 1e-9 * g["flops"]


def test_bad_layout_rejected(golden):
    g = golden("lap3d_6_llt")
    c4 = g["cblk4"].copy()
    c4[3, 3] += 1      # stride no longer equals the sum of blok heights
    la = _lib.LayoutArrays(c4, g["blok4"])
    h = ctypes.c_void_p()
    rc = _lib.lib().pastix_amd_plan_create(ctypes.byref(la.c), 0, 1, None, ctypes.byref(h))
    assert rc == -6 and not h
    rc = _lib.lib().pastix_amd_plan_create(None, 0, 1, None, ctypes.byref(h))
    assert rc == -1
