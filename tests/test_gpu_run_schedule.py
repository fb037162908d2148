"""GPU: the run schedule (one dependency-driven launch for the thin levels at the top of the tree) against the reference's
factors and against the level-by-level schedule of the same plan, which it must reproduce BIT FOR BIT (same tasks, same
piece order, same arithmetic -- only who runs when differs)."""
import os

import numpy as np
import pytest

import oracle_lib
from conftest import golden_names, recut_mask
from pastix_amd import Plan
from pastix_amd import symbolic as sy
from pastix_amd._lib import PastixAmdError

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture
def run_env():
    keep = {k: os.environ.get(k) for k in ("PASTIX_AMD_RUN", "PASTIX_AMD_RUN_TIMEOUT")}
    os.environ["PASTIX_AMD_RUN_TIMEOUT"] = "5"
    yield os.environ
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def both(p, fill, crit, env):
    out = {}
    for mode in ("0", "1"):
        env["PASTIX_AMD_RUN"] = mode
        fill()
        st = p.factorize(crit)
        out[mode] = (p.download()[0], st)
    return out


@pytest.mark.parametrize("maxc", [0, 1, 100000])
@pytest.mark.parametrize("name", golden_names("llt"))
def test_run_matches_reference_golden_and_the_level_schedule(name, maxc, golden, run_env):
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], 0, run_schedule=1, run_max_cblks=maxc) as p:
        o = both(p, lambda: p.upload(g["L0"]), g["critere"], run_env)
    L1, st = o["1"]
    scale = np.abs(g["L1"]).max()
    assert np.abs(L1 - g["L1"])[recut_mask(g["cblk4"])].max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"] == o["0"][1]["nbpivot"]
    assert np.array_equal(L1, o["0"][0])


@pytest.mark.parametrize("N,bs,maxc,dw", [(20, 32, 0, 0), (24, 128, 100000, 2), (36, 64, 16, 1), (40, 128, 0, 0)])
def test_run_is_bitwise_the_level_schedule_on_produced_layouts(N, bs, maxc, dw, run_env):
    n, cp, r, v = sy.laplacian_3d(N)
    rng = np.random.default_rng(N)
    v = v.copy()
    v[cp[:-1] - 1] *= 1.0 + 0.3 * rng.random(n)       # (the diagonal entries, first of every column: still positive definite)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=bs)
    c4, b4 = s["cblk4"], s["blok4"]
    with Plan(c4, b4, 0, run_max_cblks=maxc, run_d_workers=dw) as p:
        o = both(p, lambda: p.fill_csc(1, n, cp, r, v, s["perm"]), 1e-14, run_env)
        # and again: a refactorization through the run gives the same bits
        run_env["PASTIX_AMD_RUN"] = "1"
        p.fill_csc(1, n, cp, r, v, s["perm"])
        p.factorize(1e-14)
        L2 = p.download()[0]
    assert np.array_equal(o["0"][0], o["1"][0])
    assert np.array_equal(L2, o["1"][0])
    assert o["1"][1]["nupdate_launches"] < o["0"][1]["nupdate_launches"]
    if N <= 24:
        L0, _ = oracle_lib.fill(0, 1, n, cp, r, v, s["perm"], c4, b4)
        Lo, _, nb = oracle_lib.sopalin(0, c4, b4, L0, None, 1e-14)
        assert np.abs(o["1"][0] - Lo).max() <= TOL * np.abs(Lo).max()


def test_static_pivots_and_error_flag_through_the_run(golden, run_env):
    g = golden("lap3d_8_llt")
    c4 = g["cblk4"]
    crit = 5.9
    Lo, _, nbo = oracle_lib.sopalin(0, c4, g["blok4"], g["L0"], None, crit)
    with Plan(c4, g["blok4"], 0, run_max_cblks=100000) as p:
        run_env["PASTIX_AMD_RUN"] = "1"
        p.upload(g["L0"])
        st = p.factorize(crit)
        L1, _ = p.download()
        assert st["nbpivot"] == nbo > 0
        assert np.abs(L1 - Lo).max() <= TOL * np.abs(Lo).max()
        # not positive definite: reported, not hung
        bad = g["L0"].copy()
        bad[0] = -1.0
        p.upload(bad)
        with pytest.raises(PastixAmdError) as e:
            p.factorize(1e-300)
        assert e.value.code == -4


def test_solve_after_a_run_factorization(run_env):
    N = 24
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=64)
    import scipy.sparse as sp
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    A = A + sp.tril(A, -1).T
    x0 = np.linspace(1, 2, n)
    b = A @ x0
    with Plan(s["cblk4"], s["blok4"], 0) as p:
        run_env["PASTIX_AMD_RUN"] = "1"
        p.fill_csc(1, n, cp, r, v, s["perm"])
        p.factorize(1e-14)
        bp = np.empty(n)
        bp[s["perm"]] = b
        x = p.solve(bp)[s["perm"]]
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-10
