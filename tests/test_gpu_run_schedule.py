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
    if o["1"][1]["run_time"] > 0:             # (a run that stopped is redone level by level: bitwise the same, launch by launch)
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


def test_a_run_that_trips_its_bounded_wait_is_redone_on_the_level_schedule(run_env):
    """PASTIX_AMD_RUN_TIMEOUT far below what the resident diagonal workers wait while the levels below the run are
    factorized: the run gives up.  After fill_csc the library restores the input from the cached fill and redoes the
    factorization on the level-by-level schedule -- the factors are bitwise those of PASTIX_AMD_RUN=0 and the plan reports
    that no run launch was used; after an upload there is nothing to restore from and PASTIX_AMD_ERR_DEVICE is returned."""
    N = 40
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    with Plan(s["cblk4"], s["blok4"], 0) as p:
        run_env["PASTIX_AMD_RUN"] = "0"
        p.fill_csc(1, n, cp, r, v, s["perm"])
        L_in, _ = p.download()
        p.factorize(1e-14)
        L_level, _ = p.download()
        run_env["PASTIX_AMD_RUN"] = "1"
        p.refill()
        st = p.factorize(1e-14)
        assert st["run_tickets"] > 0                       # (this plan has a run, and it works)
        L_run, _ = p.download()
        assert np.array_equal(L_run, L_level)
        run_env["PASTIX_AMD_RUN_TIMEOUT"] = "0.00001"      # 10 us (clamped to 1 ms inside; the diagonal workers: 3 ms)
        p.refill()
        st = p.factorize(1e-14)
        L_back, _ = p.download()
        assert np.array_equal(L_back, L_level)
        assert st["run_time"] == 0.0                       # the second attempt was the level schedule
        p.upload(L_in)
        with pytest.raises(PastixAmdError) as e:
            p.factorize(1e-14)
        assert e.value.code == -3
        run_env["PASTIX_AMD_RUN_TIMEOUT"] = "5"
        p.refill()
        p.factorize(1e-14)
        assert np.array_equal(p.download()[0], L_level)


@pytest.mark.parametrize("facto", [0, 2])
def test_one_shot_call_whose_run_trips_is_redone_although_finished_panels_went_home_early(facto, run_env):
    """The one-shot entry points copy the panels of the levels below the run into the caller's buffers WHILE the run
    factorizes the rest -- over the input a stopped run would be redone from.  The input is therefore kept on the device
    for the call: with the forced expiry the call restores it from there, factorizes level by level and returns what the
    level schedule returns, bit for bit (LLt: one arena; LU: two, with re-cut cblks' blocks exchanged between them)."""
    from pastix_amd.solver import sopalin_tabs
    from pastix_amd import _lib
    N = 40
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    c4, b4 = s["cblk4"], s["blok4"]
    with Plan(c4, b4, facto) as p:
        p.fill_csc(1 if facto == 0 else 0, n, cp, r, v, s["perm"])
        L0, U0 = p.download()
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])

    def call():
        tabs = [L0[off[k]:off[k + 1]].copy() for k in range(len(w))]
        utabs = [U0[off[k]:off[k + 1]].copy() for k in range(len(w))] if facto == 2 else None
        st = sopalin_tabs(facto, c4, b4, tabs, utabs, critere=1e-14)
        return np.concatenate(tabs), (np.concatenate(utabs) if utabs else None), st

    keep_dev = os.environ.get("PASTIX_AMD_DEV")
    os.environ["PASTIX_AMD_DEV"] = "early_out_min=0"       # (early copies are made from 2e12 flop on; this problem has 3e10)
    try:
        run_env["PASTIX_AMD_RUN"] = "0"
        Ll, Ul, _ = call()
        run_env["PASTIX_AMD_RUN"] = "1"
        Lr, Ur, st = call()
        assert st["run_time"] > 0.0 and np.array_equal(Lr, Ll) and (Ul is None or np.array_equal(Ur, Ul))
        run_env["PASTIX_AMD_RUN_TIMEOUT"] = "0.00001"
        Lb, Ub, st = call()
        if st["run_time"] != 0.0:
            pytest.skip("the run finished before the forced expiry could strike")
        assert np.array_equal(Lb, Ll) and (Ul is None or np.array_equal(Ub, Ul))
    finally:
        _lib.lib().pastix_amd_release_cached_plan()
        if keep_dev is None:
            os.environ.pop("PASTIX_AMD_DEV", None)
        else:
            os.environ["PASTIX_AMD_DEV"] = keep_dev


@pytest.mark.parametrize("name", ["rlap3d_20_lu_bs128", "rlap3d_12_ldlt", "zrlap3d_20_ldlt_bs128", "zrlap3d_12_ldlh"])
def test_the_way_back_restores_every_arena(name, golden, run_env):
    """The same forced expiry for LU (two arenas), LDLt and complex LDLt / LDLh (Re / Im planes, the L D copies): the cached
    fill restores every plane and the level schedule's factors come back."""
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    cz = np.iscomplexobj(g["L0"])
    kw = {"floattype": COMPLEXDOUBLE} if cz else {}
    with Plan(g["cblk4"], g["blok4"], int(g["facto"]), run_max_cblks=100000, **kw) as p:
        run_env["PASTIX_AMD_RUN"] = "0"
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        p.factorize(g["critere"])
        ref = p.download()
        run_env["PASTIX_AMD_RUN"] = "1"
        p.refill()
        st = p.factorize(g["critere"])
        assert st["run_tickets"] > 0
        run_env["PASTIX_AMD_RUN_TIMEOUT"] = "0.00001"
        p.refill()
        st = p.factorize(g["critere"])
        if st["run_time"] != 0.0:
            pytest.skip("the run finished before the forced expiry could strike (a problem this small runs for well under a millisecond)")
        back = p.download()
    for a, b in zip(back, ref):
        if a is not None:
            assert np.array_equal(a, b)


def _lower_mask(c4):
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    parts = []
    for k in range(len(w)):
        wk, sk = int(w[k]), int(c4[k, 3])
        m = np.ones((wk, sk), dtype=bool)
        m[:, :wk] = np.triu(np.ones((wk, wk), dtype=bool))
        parts.append(m.ravel())
    return np.concatenate(parts)


@pytest.mark.parametrize("maxc", [0, 100000])
@pytest.mark.parametrize("name", golden_names("ldlt") + golden_names("lu"))
def test_run_ldlt_lu_match_reference_golden_and_the_level_schedule(name, maxc, golden, run_env):
    """LDLt (panel solve with the scaling by D and the L D copy in the second arena) and LU (two planes of update tasks,
    two solves per panel-solve ticket, the first-generation diagonal kernel as the resident worker) through the run."""
    g = golden(name)
    lu = g["facto"] == 2
    with Plan(g["cblk4"], g["blok4"], g["facto"], run_schedule=1, run_max_cblks=maxc) as p:
        out = {}
        for mode in ("0", "1"):
            run_env["PASTIX_AMD_RUN"] = mode
            p.upload(g["L0"], g["U0"] if lu else None)
            st = p.factorize(g["critere"])
            out[mode] = (p.download(), st)
    (L1, U1), st = out["1"]
    m = np.ones(L1.size, bool) if lu else _lower_mask(g["cblk4"])
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    if lu:
        assert np.abs(U1 - g["U1"]).max() <= TOL * max(scale, np.abs(g["U1"]).max())
    assert st["nbpivot"] == g["nbpivot"] == out["0"][1]["nbpivot"]
    assert st["inertia"] == out["0"][1]["inertia"]
    assert np.array_equal(L1[m], out["0"][0][0][m])
    if lu:
        assert np.array_equal(U1, out["0"][0][1])


@pytest.mark.parametrize("facto", [1, 2])
def test_run_ldlt_lu_bitwise_on_a_produced_layout(facto, run_env):
    N = 30
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    c4, b4 = s["cblk4"], s["blok4"]
    if facto == 2:
        import scipy.sparse as sp
        A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
        Af = (A + sp.tril(A, -1).T).tocsc()
        Af = (Af + sp.triu(Af, 1).multiply(0.2)).tocsc()
        Af.sort_indices()
        cp, r, v = Af.indptr.astype(np.int64) + 1, Af.indices.astype(np.int64) + 1, Af.data.copy()
    with Plan(c4, b4, facto, run_schedule=1) as p:
        out = {}
        for mode in ("0", "1"):
            run_env["PASTIX_AMD_RUN"] = mode
            p.fill_csc(0 if facto == 2 else 1, n, cp, r, v, s["perm"])
            st = p.factorize(1e-14)
            out[mode] = (p.download(), st)
    assert out["1"][1]["run_tickets"] > 0 and out["0"][1]["run_tickets"] == 0
    m = np.ones(out["1"][0][0].size, bool) if facto == 2 else _lower_mask(c4)
    assert np.array_equal(out["0"][0][0][m], out["1"][0][0][m])
    if facto == 2:
        assert np.array_equal(out["0"][0][1], out["1"][0][1])


@pytest.mark.parametrize("maxc", [0, 100000])
@pytest.mark.parametrize("name", golden_names("ldlt", prec="z") + golden_names("ldlh", prec="z"))
def test_run_complex_ldlt_ldlh_match_reference_golden_and_the_level_schedule(name, maxc, golden, run_env):
    """complex LDLt / LDLh through the run: update tasks on the real and imaginary planes of a tile are two chains, the panel
    solve is the parked complex solve (64 rows per ticket), the diagonal workers run k_diag_zsy_w's body."""
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE, run_schedule=1, run_max_cblks=maxc) as p:
        out = {}
        for mode in ("0", "1"):
            run_env["PASTIX_AMD_RUN"] = mode
            p.upload(g["L0"])
            st = p.factorize(g["critere"])
            out[mode] = (p.download()[0], st)
        assert out["1"][1]["run_tickets"] > 0 and out["0"][1]["run_tickets"] == 0
    L1, st = out["1"]
    m = _lower_mask(g["cblk4"])
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"] == out["0"][1]["nbpivot"]
    # (not bit for bit here: the two panel-solve kernels are different code around the same operations -- the complex
    # scaling by 1 / d is contracted into fused multiply-adds differently -- so the factors agree to rounding)
    assert np.abs(L1 - out["0"][0])[m].max() <= 1e-14 * scale, np.abs(L1 - out["0"][0])[m].max() / scale



# ---- the diagonal-blok tickets of the run launch, every width -------------------------------------------------------------
# In the run launch a real LLt / LDLt diagonal blok is a ticket of k_run_update: the bodies of k_diag_llt_w / _ldlt_w inside
# that kernel's 64 VGPRs (LU and complex bloks: a resident kernel of their own).  Which instructions a wave of such a ticket
# executes depends on the blok's width only, so the widths 1 .. 128 are ALL its paths: each must give, bit for bit, what
# the level-by-level schedule gives with the kernels' own launches.
def _dense_two(w, facto, cplx):
    from test_gpu_edges import dense_layout, panels_of, spd
    c4, b4, n = dense_layout([24, w])
    A = spd(n, 1000 + w)
    if facto == 2:
        A = A + np.triu(np.random.default_rng(w).standard_normal((n, n)), 1) * 0.1
    if cplx:
        rng = np.random.default_rng(2000 + w)
        B = rng.standard_normal((n, n)) * 0.2
        A = A + 1j * ((B + B.T) if facto == 1 else (B - B.T))      # complex symmetric (LDLt) / Hermitian (LDLh)
        if facto == 3:
            A[np.diag_indices(n)] = A[np.diag_indices(n)].real
    return c4, b4, panels_of(A, c4), (panels_of(A.T, c4) if facto == 2 else None)


@pytest.mark.parametrize("facto", [0, 1, 2])
def test_diagonal_tickets_of_every_width(facto, run_env):
    for w in range(1, 129):
        c4, b4, L0, U0 = _dense_two(w, facto, False)
        with Plan(c4, b4, facto, run_schedule=1, run_max_cblks=100000) as p:
            out = {}
            for mode in ("0", "1"):
                run_env["PASTIX_AMD_RUN"] = mode
                p.upload(L0, U0)
                st = p.factorize(1e-30)
                out[mode] = (p.download(), st)
        assert out["1"][1]["run_tickets"] > 0 and out["0"][1]["run_tickets"] == 0, w
        m = np.ones(L0.size, bool) if facto == 2 else _lower_mask(c4)
        assert np.array_equal(out["0"][0][0][m], out["1"][0][0][m]), w
        if facto == 2:
            assert np.array_equal(out["0"][0][1], out["1"][0][1]), w


def _edges_digest(p):
    import ctypes
    from pastix_amd import _lib
    out = (ctypes.c_int64 * 4)()
    rc = _lib.lib().pastix_amd_plan_run_edges_digest(p._h, out)
    assert rc == 0, rc
    return tuple(out)


@pytest.mark.parametrize("N,bs,facto", [(24, 64, 0), (40, 128, 0), (32, 128, 1), (28, 64, 2), (60, 128, 0)])
def test_reader_lists_built_on_the_device_equal_the_host_built_ones(N, bs, facto, run_env):
    """Round 6: the reader lists of the run (which update tickets read which solved 128-row tile, and every ticket's
    initial counter) are built on the GPU from the uploaded tables (csrc/run_edges.hip: key = tile << 32 | ticket, radix
    sort, unique).  PASTIX_AMD_DEV=run_host_edges builds them on host threads as rounds 4-5 did: same pairs, same counters,
    same ready set -- and bitwise the same factors (LU: the tables only, the symmetric fill does not apply)."""
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=bs)
    c4, b4 = s["cblk4"], s["blok4"]
    res = {}
    keep = os.environ.get("PASTIX_AMD_DEV")
    try:
        for mode in ("device", "host"):
            if mode == "host":
                os.environ["PASTIX_AMD_DEV"] = "run_host_edges"
            else:
                os.environ.pop("PASTIX_AMD_DEV", None)
            with Plan(c4, b4, facto, run_schedule=1) as p:
                dig = _edges_digest(p)
                if facto != 2:
                    p.fill_csc(1, n, cp, r, v, s["perm"])
                    st = p.factorize(1e-14)
                    res[mode] = (dig, p.download()[0], st["run_time"] > 0)
                else:
                    res[mode] = (dig, None, True)
    finally:
        if keep is None:
            os.environ.pop("PASTIX_AMD_DEV", None)
        else:
            os.environ["PASTIX_AMD_DEV"] = keep
    assert res["device"][0] == res["host"][0], (res["device"][0], res["host"][0])
    assert res["device"][0][0] > 0 and res["device"][0][3] > 0
    if facto != 2:
        assert np.array_equal(res["device"][1], res["host"][1])
