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


def test_bench_refuses_to_mislabel_the_gpu_count():
    """bench.py --gpus N: N ranks run or it exits non-zero BEFORE printing a line -- never `n_gpus` != ranks that ran."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # (a) launcher's world size disagrees with --gpus
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout
    # (b) no launcher and fewer GPUs than ranks (this container has none): refuses instead of running one rank
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout
