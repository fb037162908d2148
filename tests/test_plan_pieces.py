"""Host only: the update pieces of the plan -- rectangles and GATHERED pieces (options.gather_min: what one source cblk
contributes to one target tile as ONE piece with two row maps, for layouts whose bloks are fragments) -- against the
reference's definition of the update (compute_1dgemm, sopalin_compute.c:865-1032): every product (rows of blok j) x (rows
of blok i)^T, j >= i, of every source cblk is subtracted exactly once, where add_contrib_local puts it (:427-429).
pastix_amd_plan_check_pieces decodes every piece into (target entry, source row of A, source row of B) triples and
compares them with the list made from the layout alone."""
import ctypes

import numpy as np
import pytest

from conftest import golden_names
from pastix_amd import _lib
from pastix_amd import symbolic as sy
from pastix_amd._lib import LayoutArrays, Options


def check(c4, b4, facto, gather_min, **kw):
    la = LayoutArrays(c4, b4)
    o = Options()
    o.gather_min = gather_min
    for k, v in kw.items():
        setattr(o, k, v)
    out = (ctypes.c_int64 * 4)()
    rc = _lib.lib().pastix_amd_plan_check_pieces(ctypes.byref(la.c), facto, ctypes.byref(o), out)
    assert rc == 0
    return list(out)


@pytest.mark.parametrize("name", [n for n in golden_names("llt") + golden_names("ldlt")])
def test_pieces_are_the_references_products_on_blend_layouts(name, golden):
    g = golden(name)
    if len(g["cblk4"]) > 4000:
        pytest.skip("large")
    ref = check(g["cblk4"], g["blok4"], g["facto"], -1)          # rectangles only
    assert ref[0] == ref[1] and ref[2] == 0 and ref[3] == 0
    for gm in (0, 2, 7):
        o = check(g["cblk4"], g["blok4"], g["facto"], gm)
        assert o[0] == o[1] == ref[0] and o[2] == 0, (gm, o)
    # the reference's blend on the harness's separator numbering gives fragmented bloks: the default gathers them
    if name.startswith(("lap3d", "rlap3d")):
        assert check(g["cblk4"], g["blok4"], g["facto"], 0)[3] > 0


@pytest.mark.parametrize("N,bs", [(10, 32), (16, 128), (20, 64)])
def test_pieces_on_produced_layouts(N, bs):
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=bs)
    for gm in (-1, 0, 2):
        for kw in ({}, {"run_schedule": -1}, {"quadrant_min": 1, "quadrant_fill_pct": 200}):
            o = check(s["cblk4"], s["blok4"], 0, gm, **kw)
            assert o[0] == o[1] and o[2] == 0, (gm, kw, o)


def test_plan_profile_recuts_cblks_wider_than_the_panel_kernels_take():
    """pastix_amd_plan_profile (host only) on a layout with a 590-column cblk (blend's split rule leaves the root whole):
    the cblk is re-cut into column groups as pastix_amd_plan_create does -- it used to return UNSUPPORTED, which kept the
    schedule statistics (and PASTIX_AMD_DEV=mode_stats) away from the reference's own layouts."""
    from pastix_amd import dist as pd
    N = 24
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=512, blend_split=True)
    c = s["cblk4"]
    assert (c[:-1, 1] - c[:-1, 0] + 1).max() > 256
    sf, sm, stn, pf, uf = pd.plan_profile(s["cblk4"], s["blok4"], None, 0)
    assert len(sf) > 1 and sf.sum() > 0 and pf.sum() > 0 and stn.sum() > 0
