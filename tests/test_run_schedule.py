"""The run schedule (plan.h RunInfo: the thin levels at the top of the tree as one dependency-driven launch, tasks gated
by counters the way the reference's tasks wait for TASK_CTRBCNT, sopalin3d.c:790-1025): host-side checks.
pastix_amd_plan_run_info builds the plan without a device and REPLAYS the counter protocol: every ticket and diagonal
task must run, and must find at its turn what it needs (its tile written exactly `seq` times, its source tiles solved,
its diagonal blok factorized)."""
import ctypes

import numpy as np
import pytest

from conftest import golden_names
from pastix_amd import _lib
from pastix_amd import symbolic as sy
from pastix_amd.solver import LayoutArrays


def run_info(c4, b4, facto=0, floattype=1, **kw):
    la = LayoutArrays(c4, b4)
    o = _lib.Options()
    for k, v in kw.items():
        setattr(o, k, v)
    info = (ctypes.c_int64 * 8)()
    rc = _lib.lib().pastix_amd_plan_run_info(ctypes.byref(la.c), facto, floattype, ctypes.byref(o), info)
    assert rc == 0
    return dict(zip(("L0", "levels", "tickets", "edges", "dworkers", "solves", "flops", "verify"), list(info)))


@pytest.mark.parametrize("name", golden_names("llt"))
def test_replay_on_the_reference_layouts(name, golden):
    g = golden(name)
    for kw in ({}, {"run_max_cblks": 1}, {"run_max_cblks": 1000, "run_d_workers": 3}):
        r = run_info(g["cblk4"], g["blok4"], 0, **kw)
        assert r["verify"] == 0, (name, kw, r)
        if r["L0"] >= 0:
            assert r["levels"] - r["L0"] >= 2 and r["tickets"] > 0 and r["solves"] > 0


@pytest.mark.parametrize("N,bs", [(10, 16), (16, 32), (24, 128), (30, 64)])
def test_replay_on_produced_layouts(N, bs):
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=bs)
    for kw in ({}, {"run_max_cblks": 4}, {"run_max_cblks": 100000}):
        ri = run_info(s["cblk4"], s["blok4"], 0, **kw)
        assert ri["verify"] == 0, (N, bs, kw, ri)
        assert ri["L0"] >= 0
    # everything in the run: its flops are the update flops of the whole plan but the leaf level's own panels
    assert run_info(s["cblk4"], s["blok4"], 0, run_max_cblks=100000)["L0"] == 0


def test_run_can_be_switched_off_and_covers_the_real_factorizations():
    n, cp, r, v = sy.laplacian_3d(12)
    perm, _ = sy.order_grid(12, 12, 12)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=32)
    c4, b4 = s["cblk4"], s["blok4"]
    assert run_info(c4, b4, 0, run_schedule=-1)["L0"] == -1
    llt = run_info(c4, b4, 0)
    assert llt["L0"] >= 0
    for facto in (1, 2):                       # LDLt: the same tickets; LU: a second plane of update tasks
        ri = run_info(c4, b4, facto)
        assert ri["L0"] == llt["L0"] and ri["verify"] == 0 and ri["solves"] == llt["solves"]
        assert ri["tickets"] == llt["tickets"] if facto == 1 else ri["tickets"] > llt["tickets"]


@pytest.mark.parametrize("name", golden_names("ldlt") + golden_names("lu"))
def test_replay_on_the_reference_layouts_ldlt_lu(name, golden):
    g = golden(name)
    if (g["cblk4"][:-1, 1] - g["cblk4"][:-1, 0] + 1).max() > 256:
        pytest.skip("cblks wider than 256 columns reach the planner re-cut (api.cpp build_split): covered on the GPU")
    for kw in ({"run_schedule": 1}, {"run_schedule": 1, "run_max_cblks": 1000, "run_d_workers": 2}):
        r = run_info(g["cblk4"], g["blok4"], g["facto"], **kw)
        assert r["verify"] == 0, (name, kw, r)


def test_schur_layout_keeps_its_last_cblk_out_of_the_diagonal_tasks(golden):
    g = golden("rlap3d_12_llt")
    r = run_info(g["cblk4"], g["blok4"], 0, schur=1)
    assert r["verify"] == 0


@pytest.mark.parametrize("name", golden_names("ldlt", prec="z") + golden_names("ldlh", prec="z"))
def test_replay_on_the_reference_layouts_complex(name, golden):
    """complex LDLt / LDLh: update tasks on the real and imaginary planes of a tile are two chains; a panel solve is two
    64-row tickets per 128-row tile."""
    g = golden(name)
    if (g["cblk4"][:-1, 1] - g["cblk4"][:-1, 0] + 1).max() > 256:
        pytest.skip("re-cut layouts: covered on the GPU")
    for kw in ({}, {"run_max_cblks": 1000, "run_d_workers": 2}):
        r = run_info(g["cblk4"], g["blok4"], g["facto"], floattype=3, **kw)
        assert r["verify"] == 0, (name, kw, r)

