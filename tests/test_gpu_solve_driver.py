"""GPU: device fill/refill, device solves, the pastix()/iparm/dparm entry point, and
size-independent properties at sizes the oracle cannot reach."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib
from pastix_amd import Plan
from pastix_amd import pastix as px
from pastix_amd import symbolic as sy

pytestmark = pytest.mark.gpu


def _sym_matvec(n, cp, r, v):
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    return A + sp.tril(A, -1).T


def test_solve_matches_oracle(golden):
    g = golden("rlap3d_12_llt")
    bp = np.empty_like(g["b"])
    bp[g["perm"]] = g["b"]
    with Plan(g["cblk4"], g["blok4"], 0) as p:
        p.upload(g["L0"])
        p.factorize(g["critere"])
        x = p.solve(bp)[g["perm"]]
    assert np.abs(x - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()


@pytest.mark.parametrize("name", ["rlap3d_12_ldlt", "rlap3d_12_lu", "rlap3d_8_lu"])
def test_solve_ldlt_lu_matches_reference(name, golden):
    g = golden(name)
    bp = np.empty_like(g["b"])
    bp[g["perm"]] = g["b"]
    with Plan(g["cblk4"], g["blok4"], g["facto"]) as p:
        p.upload(g["L0"], g["U0"])
        p.factorize(g["critere"])
        x = p.solve(bp)[g["perm"]]
    assert np.abs(x - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()


@pytest.mark.parametrize("name", ["rlap3d_12_llt", "rlap3d_12_ldlt", "rlap3d_12_lu"])
@pytest.mark.parametrize("nrhs", [2, 4, 7])
def test_multi_rhs_solve_equals_single_rhs_solves(name, nrhs, golden):
    """Several right-hand sides share one pass over the panels (groups of 4, 2, 1): every column must equal the
    single-vector solve, and column 0 the reference's solution."""
    g = golden(name)
    n = len(g["b"])
    rng = np.random.default_rng(nrhs)
    B = np.empty((n, nrhs))
    B[:, 0] = g["b"]
    B[:, 1:] = rng.standard_normal((n, nrhs - 1))
    Bp = np.empty_like(B)
    Bp[g["perm"]] = B
    with Plan(g["cblk4"], g["blok4"], g["facto"]) as p:
        p.upload(g["L0"], g.get("U0"))
        p.factorize(g["critere"])
        X = p.solve(Bp.copy())
        singles = np.stack([p.solve(Bp[:, j].copy()) for j in range(nrhs)], axis=1)
    scale = np.abs(singles).max()
    assert np.abs(X - singles).max() <= 1e-13 * scale
    assert np.abs(X[g["perm"], 0] - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()


def test_refill_is_idempotent_and_refactor_is_deterministic(golden):
    g = golden("rlap3d_10_llt")
    with Plan(g["cblk4"], g["blok4"], 0) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        p.factorize(g["critere"])
        La, _ = p.download()
        p.refill()
        L0, _ = p.download()
        assert np.array_equal(L0, g["L0"])
        p.factorize(g["critere"])
        Lb, _ = p.download()
    assert np.array_equal(La, Lb)          # tile ownership => bitwise reproducible


# (36, 128): remainder cblks 8 and 16 columns wide feed whole 128x128 tiles -> the K-tail of the DMA loop
@pytest.mark.parametrize("N,bs", [(24, 128), (32, 64), (36, 128), (40, 256)])
def test_own_layout_vs_oracle_and_residual(N, bs):
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=bs)
    c4, b4 = s["cblk4"], s["blok4"]
    with Plan(c4, b4, 0) as p:
        p.fill_csc(1, n, cp, r, v, s["perm"])
        st = p.factorize(1e-14)
        L1, _ = p.download()
        b = np.random.default_rng(3).random(n)
        bp = np.empty(n)
        bp[s["perm"]] = b
        x = p.solve(bp)[s["perm"]]
    assert st["nbpivot"] == 0
    A = _sym_matvec(n, cp, r, v)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-10      # SURVEY 8d end-to-end bar
    if N <= 36:
        L0, _ = oracle_lib.fill(0, 1, n, cp, r, v, s["perm"], c4, b4)
        Lo, _, _ = oracle_lib.sopalin(0, c4, b4, L0, None, 1e-14)
        assert np.abs(L1 - Lo).max() <= 1e-12 * np.abs(Lo).max()


def test_pastix_entry_point_config1():
    """BASELINE config 1: the reference's -lap 1000 generator (1-D, diag 2, sub-diagonal -1,
    laplacian.c:151-181), rhs e_1 + e_n, double LLt through pastix()/iparm/dparm."""
    n = 1000
    colptr = np.arange(1, 2 * n + 1, 2, dtype=np.int64)
    colptr = np.concatenate([colptr[:n], [2 * n]]).astype(np.int64)
    rows = np.empty(2 * n - 1, dtype=np.int64)
    vals = np.empty(2 * n - 1)
    rows[0::2] = np.arange(1, n + 1)
    vals[0::2] = 2.0
    rows[1::2] = np.arange(2, n + 1)
    vals[1::2] = -1.0
    b = np.zeros(n)
    b[0] = b[-1] = 1.0
    rhs = b.copy()
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    iparm, dparm = px.init_param()
    assert iparm[px.IPARM["MAX_BLOCKSIZE"]] == 120 and dparm[px.DPARM["EPSILON_MAGN_CTRL"]] == 1e-31
    iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LLT
    iparm[px.IPARM["SYM"]] = px.API_SYM_YES
    iparm[px.IPARM["START_TASK"]] = px.API_TASK["ORDERING"]
    iparm[px.IPARM["END_TASK"]] = px.API_TASK["REFINE"]
    pd = px.pastix(None, n, colptr, rows, vals, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0
    A = _sym_matvec(n, colptr, rows, vals)
    assert np.linalg.norm(A @ b - rhs) / np.linalg.norm(rhs) <= 1e-12
    assert dparm[px.DPARM["RELATIVE_ERROR"]] <= 1e-12 and dparm[px.DPARM["FACT_TIME"]] > 0
    assert iparm[px.IPARM["STATIC_PIVOTING"]] == 0 and iparm[px.IPARM["NNZEROS"]] > 0
    iparm[px.IPARM["START_TASK"]] = iparm[px.IPARM["END_TASK"]] = px.API_TASK["CLEAN"]
    px.pastix(pd, n, colptr, rows, vals, perm, invp, b, 1, iparm, dparm)


def test_pastix_personal_ordering_grid():
    N = 12
    n, cp, r, v = sy.laplacian_3d(N)
    perm0, invp0 = sy.order_grid(N, N, N)
    perm, invp = perm0 + 1, invp0 + 1            # same base as the CSC (kass.c:143-156)
    b = np.random.default_rng(5).random(n)
    rhs = b.copy()
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LLT
    iparm[px.IPARM["ORDERING"]] = px.API_ORDER_PERSONAL
    iparm[px.IPARM["END_TASK"]] = px.API_TASK["SOLVE"]
    pd = px.pastix(None, n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0
    A = _sym_matvec(n, cp, r, v)
    assert np.linalg.norm(A @ b - rhs) / np.linalg.norm(rhs) < 1e-12
    iparm[px.IPARM["START_TASK"]] = iparm[px.IPARM["END_TASK"]] = px.API_TASK["CLEAN"]
    px.pastix(pd, n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_plans_emulated_on_one_gpu(world, golden):
    """The per-rank plans of the multi-GPU path (ownership, shadow panels, level-stepped API) run
    on ONE device with an in-process exchange; result must equal the reference factors."""
    import torch
    from pastix_amd import dist as pd
    g = golden("rlap3d_14_llt_bs24")
    c4, b4 = g["cblk4"], g["blok4"]
    owner = pd.partition(c4, b4, world, split=2)
    level = pd.levels_of(c4, b4)
    engs = [pd.GpuEngine(c4, b4, owner, r, 0) for r in range(world)]
    exch = [pd.Exchange(c4, b4, owner, level, r) for r in range(world)]
    assert sum(len(x) for e in exch for x in e.sends) > 0
    # a caller-owned arena smaller than pastix_amd_plan_arena_info asks for is refused (the DMA slack is checked,
    # not a convention)
    import ctypes
    from pastix_amd import _lib
    ne, first = ctypes.c_int64(0), ctypes.c_int64(0)
    assert _lib.lib().pastix_amd_plan_arena_info(engs[0]._h, ctypes.byref(ne), ctypes.byref(first)) == 0
    assert first.value > 0 and ne.value >= int(engs[0].poff[-1]) + 2 * first.value
    assert _lib.lib().pastix_amd_plan_set_arena(engs[0]._h, ctypes.c_void_p(engs[0]._arena_store.data_ptr()), None,
                                                ctypes.c_int64(ne.value - 1)) == -1      # PASTIX_AMD_ERR_BADPARAMETER
    for e in engs:
        assert np.array_equal(e.level, level)
        e.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        e.begin(g["critere"])
    for l in range(exch[0].nlevels):
        for e in engs:
            e.update(l)
        for r in range(world):
            for t, dst in exch[r].sends[l]:
                assert engs[r].panel(t).numel() == engs[dst].recv_numel(t, r)
                engs[dst].add(t, engs[r].panel(t).clone(), r)
        for e in engs:
            e.panels(l)
    nb = sum(e.end()["nbpivot"] for e in engs)
    assert nb == g["nbpivot"]
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    scale = np.abs(g["L1"]).max()
    for k in range(len(w)):
        got = engs[owner[k]].panel(k).cpu().numpy()
        assert np.abs(got - g["L1"][off[k]:off[k + 1]]).max() <= 1e-12 * scale
    for e in engs:
        e.close()


@pytest.mark.parametrize("N", [5, 8])
def test_config5_z_ldlt_elasticity_pattern(N):
    """BASELINE config 5: complex double LDLt on a 3-dof elasticity-pattern matrix (complex symmetric),
    GPU (split-plane zgemm on MFMA) vs the CPU oracle; tolerance 1e-12 * max|L|."""
    from pastix_amd import COMPLEXDOUBLE
    n, cp, r, v, _ = sy.elasticity_3d(N)
    perm, _ = sy.order_grid_dof(N, 3)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=96)
    c4, b4 = s["cblk4"], s["blok4"]
    L0, _ = oracle_lib.fill(1, 1, n, cp, r, v, s["perm"], c4, b4)
    crit = 1e-12
    Lo, _, nbo = oracle_lib.sopalin(1, c4, b4, L0, None, crit)
    with Plan(c4, b4, 1, floattype=COMPLEXDOUBLE) as p:
        p.fill_csc(1, n, cp, r, v, s["perm"])
        Ld, _ = p.download()
        assert np.array_equal(Ld, L0)
        st = p.factorize(crit)
        L1, _ = p.download()
        # end-to-end: complex triangular solves on the device, against the oracle's substitution
        b = np.random.default_rng(2).random(n) + 1j * np.random.default_rng(3).random(n)
        bp = np.empty(n, dtype=np.complex128)
        bp[s["perm"]] = b
        xd = p.solve(bp.copy())
    assert st["nbpivot"] == nbo == 0
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    m = np.ones(len(L1), dtype=bool)
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    for k in range(len(w)):
        for c in range(int(w[k])):
            m[off[k] + c * c4[k, 3]: off[k] + c * c4[k, 3] + c] = False
    assert np.abs(L1 - Lo)[m].max() <= 1e-12 * np.abs(Lo[m]).max()
    xo = oracle_lib.solve(1, c4, b4, L1, None, bp)
    assert np.abs(xd - xo).max() <= 1e-11 * np.abs(xo).max()
    x = xd[s["perm"]]
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    A = A + sp.tril(A, -1).T
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-10


def _run_tasks(pd, first, last, n, cp, r, v, perm, invp, b, nrhs, iparm, dparm):
    iparm[px.IPARM["START_TASK"]] = px.API_TASK[first]
    iparm[px.IPARM["END_TASK"]] = px.API_TASK[last]
    pd = px.pastix(pd, n, cp, r, v, perm, invp, b, nrhs, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0
    return pd


def test_pastix_step_by_step_multi_rhs_and_refactorization():
    """The call sequences of the reference's examples step-by-step.c and multi-rhs.c: every task on its own,
    several right-hand sides in one solve, then new values with the analysis kept (NUMFACT .. REFINE again)."""
    N = 10
    n, cp, r, v = sy.laplacian_3d(N)
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LLT
    pd = px.PastixData()
    pd.set_grid(N, N, N)
    nrhs = 3
    B = np.asfortranarray(np.random.default_rng(8).random((n, nrhs)))
    b = B.reshape(-1, order="F").copy()
    for task in ("ORDERING", "SYMBFACT", "ANALYSE", "NUMFACT", "SOLVE", "REFINE"):
        pd = _run_tasks(pd, task, task, n, cp, r, v, perm, invp, b, nrhs, iparm, dparm)
    A = _sym_matvec(n, cp, r, v)
    X = b.reshape(n, nrhs, order="F")
    assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-12
    assert sorted(perm.tolist()) == list(range(1, n + 1)) and np.array_equal(perm[invp - 1], np.arange(1, n + 1))
    # second factorization on the same analysis with other values
    v2 = v.copy()
    v2[v2 > 0] *= 1.5
    b2 = np.random.default_rng(9).random(n)
    rhs2 = b2.copy()
    pd = _run_tasks(pd, "NUMFACT", "REFINE", n, cp, r, v2, perm, invp, b2, 1, iparm, dparm)
    A2 = _sym_matvec(n, cp, r, v2)
    assert np.linalg.norm(A2 @ b2 - rhs2) / np.linalg.norm(rhs2) < 1e-12
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v2, perm, invp, b2, 1, iparm, dparm)


def test_pastix_reentrant_two_instances():
    """reentrant.c: two independent pastix_data handles alive at once, tasks interleaved."""
    out = []
    for N, seed in ((8, 1), (9, 2)):
        n, cp, r, v = sy.laplacian_3d(N)
        iparm, dparm = px.init_param()
        iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LLT
        pd = px.PastixData()
        pd.set_grid(N, N, N)
        b = np.random.default_rng(seed).random(n)
        out.append(dict(N=N, n=n, cp=cp, r=r, v=v, iparm=iparm, dparm=dparm, pd=pd, b=b, rhs=b.copy(),
                        perm=np.zeros(n, dtype=np.int64), invp=np.zeros(n, dtype=np.int64)))
    for first, last in (("ORDERING", "ANALYSE"), ("NUMFACT", "NUMFACT"), ("SOLVE", "SOLVE")):
        for o in out:
            o["pd"] = _run_tasks(o["pd"], first, last, o["n"], o["cp"], o["r"], o["v"], o["perm"], o["invp"], o["b"], 1,
                                 o["iparm"], o["dparm"])
    for o in out:
        A = _sym_matvec(o["n"], o["cp"], o["r"], o["v"])
        assert np.linalg.norm(A @ o["b"] - o["rhs"]) / np.linalg.norm(o["rhs"]) < 1e-12
        _run_tasks(o["pd"], "CLEAN", "CLEAN", o["n"], o["cp"], o["r"], o["v"], o["perm"], o["invp"], o["b"], 1,
                   o["iparm"], o["dparm"])


@pytest.mark.parametrize("facto", ["LDLT", "LU"])
def test_pastix_ldlt_and_lu_through_the_entry_point(facto):
    N = 9
    full = facto == "LU"
    n, cp, r, v = sy.laplacian_3d(N, full=full)
    if full:                                   # unsymmetric values on the symmetric pattern
        v = v * (1.0 + 0.1 * np.random.default_rng(4).random(len(v)))
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LU if full else px.API_FACT_LDLT
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if full else px.API_SYM_YES
    pd = px.PastixData()
    pd.set_grid(N, N, N)
    b = np.random.default_rng(6).random(n)
    rhs = b.copy()
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    pd = _run_tasks(pd, "ORDERING", "REFINE", n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    import scipy.sparse as sp
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    if not full:
        A = A + sp.tril(A, -1).T
        assert iparm[px.IPARM["INERTIA"]] == n            # SPD: every D entry positive (sopalin3d.c:1144-1160)
    assert np.linalg.norm(A @ b - rhs) / np.linalg.norm(rhs) < 1e-12
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("mode", ["GMRES", "GRAD", "PIVOT", "BICGSTAB"])
def test_pastix_refinement_modes_recover_from_static_pivoting(mode):
    """IPARM_REFINEMENT (api.h:353-365).  A huge DPARM_EPSILON_MAGN_CTRL makes the factorization clamp pivots
    (static pivoting perturbs the factors), so the first solve is inaccurate and the refiner has real work."""
    N = 8
    n, cp, r, v = sy.laplacian_3d(N)
    iparm, dparm = px.init_param()
    assert iparm[px.IPARM["REFINEMENT"]] == px.API_RAF_GMRES and iparm[px.IPARM["GMRES_IM"]] == 25
    iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LLT
    iparm[px.IPARM["REFINEMENT"]] = getattr(px, "API_RAF_" + mode)
    dparm[px.DPARM["EPSILON_MAGN_CTRL"]] = -5.9          # critere = 5.9: some pivots of this matrix are clamped
    pd = px.PastixData()
    pd.set_grid(N, N, N)
    b = np.random.default_rng(12).random(n)
    rhs = b.copy()
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    pd = _run_tasks(pd, "ORDERING", "SOLVE", n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["STATIC_PIVOTING"]] > 0
    A = _sym_matvec(n, cp, r, v)
    assert np.linalg.norm(A @ b - rhs) / np.linalg.norm(rhs) > 1e-6       # perturbed factors: poor first solve
    pd = _run_tasks(pd, "REFINE", "REFINE", n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert np.linalg.norm(A @ b - rhs) / np.linalg.norm(rhs) < 1e-11
    assert 0 < iparm[px.IPARM["NBITER"]] <= 250 and dparm[px.DPARM["RELATIVE_ERROR"]] < 1e-11
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("name", ["zrlap3d_8_ldlt", "zrlap3d_12_ldlt", "zrlap3d_8_ldlh", "zrlap3d_12_ldlh", "zrlap3d_8_lu",
                                  "zrlap3d_12_lu", "zyoung4c_841_ldlt"])
def test_z_solve_matches_reference(name, golden):
    """Complex device solves (LDLt, LDLh, LU) on the device-resident factors vs the reference's own solution."""
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        p.factorize(g["critere"])
        bp = np.empty(g["n"], dtype=np.complex128)
        bp[g["perm"]] = g["b"]
        x = p.solve(bp.copy())[g["perm"]]                       # (solve works in place, like the reference's b)
        X2 = p.solve(np.stack([bp, 2j * bp], axis=1))           # two right-hand sides at once
    assert np.abs(x - g["x"]).max() <= 1e-10 * np.abs(g["x"]).max()
    assert np.abs(X2[:, 1][g["perm"]] - 2j * g["x"]).max() <= 1e-10 * 2 * np.abs(g["x"]).max()


@pytest.mark.parametrize("facto,ns", [("LLT", 40), ("LDLT", 36), ("LU", 300)])
def test_pastix_schur_mode(facto, ns):
    """IPARM_SCHUR through the entry point (the reference's schur.c example): pastix_setSchurUnknownList isolates
    unknowns at the end, the factorization leaves their dense Schur complement in the last cblk, pastix_getSchur
    returns it; checked against S = A22 - A21 A11^-1 A12 computed densely."""
    import scipy.sparse as sp
    N = 9
    full = facto == "LU"
    n, cp, r, v = sy.laplacian_3d(N, full=full)
    if full:
        v = v * (1.0 + 0.1 * np.random.default_rng(4).random(len(v)))
    rng = np.random.default_rng(21)
    schur = np.sort(rng.choice(n, size=ns, replace=False)) + 1          # 1-based like the CSC
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = getattr(px, "API_FACT_" + facto)
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if full else px.API_SYM_YES
    iparm[px.IPARM["SCHUR"]] = px.API_YES
    pd = px.PastixData()
    pd.set_grid(N, N, N)
    pd.set_schur_unknown_list(schur)
    b = np.zeros(n)
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    pd = _run_tasks(pd, "ORDERING", "NUMFACT", n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert sorted((invp[n - ns:]).tolist()) == schur.tolist()             # the listed unknowns come last
    S = pd.get_schur(ns)
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n)).toarray()
    if not full:
        A = A + np.tril(A, -1).T
    order = invp - 1                                                      # new -> old
    Ap = A[np.ix_(order, order)]
    n1 = n - ns
    Sref = Ap[n1:, n1:] - Ap[n1:, :n1] @ np.linalg.solve(Ap[:n1, :n1], Ap[:n1, n1:])
    if full:
        assert np.abs(S - Sref).max() <= 1e-10 * np.abs(Sref).max()
    else:
        tri = np.tril_indices(ns)
        assert np.abs(S - Sref)[tri].max() <= 1e-10 * np.abs(Sref).max()
    # solves are not part of Schur mode here
    iparm[px.IPARM["START_TASK"]] = iparm[px.IPARM["END_TASK"]] = px.API_TASK["SOLVE"]
    px.pastix(pd, n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == -5
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("facto,herm", [("LDLT", False), ("LDLH", True), ("LU", False)])
def test_z_pastix_entry_point(facto, herm):
    """Z_pastix: complex double through the entry point -- BASELINE config 5 (complex-symmetric LDLt on the 3-dof
    elasticity pattern), Hermitian LDLh and complex LU; solve + GMRES/CG refinement in complex arithmetic."""
    import scipy.sparse as sp
    N = 6
    n, cp, r, v, _ = sy.elasticity_3d(N)                     # complex symmetric, lower triangle, diagonally dominant
    rng = np.random.default_rng(17)
    if herm:                                                  # Hermitian: real diagonal (already), lower triangle as given
        pass
    full = facto == "LU"
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    low = sp.tril(A, -1)
    Afull = (A + (low.conj().T if herm else low.T)).tocsc()
    if full:
        Afull = Afull + sp.triu(Afull, 1).multiply(0.05)      # unsymmetric values on the symmetric pattern
        Afull = Afull.tocsc()
        Afull.sort_indices()
        cp, r, v = Afull.indptr.astype(np.int64) + 1, Afull.indices.astype(np.int64) + 1, Afull.data.astype(np.complex128)
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FLOAT"]] = px.API_COMPLEXDOUBLE
    iparm[px.IPARM["FACTORIZATION"]] = getattr(px, "API_FACT_" + facto)
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if full else (px.API_SYM_HER if herm else px.API_SYM_YES)
    iparm[px.IPARM["REFINEMENT"]] = px.API_RAF_GRAD if herm else px.API_RAF_GMRES
    b = (rng.random(n) + 1j * rng.random(n)).astype(np.complex128)
    rhs = b.copy()
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    pd = _run_tasks(None, "ORDERING", "REFINE", n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert np.linalg.norm(Afull @ b - rhs) / np.linalg.norm(rhs) < 1e-11
    assert dparm[px.DPARM["RELATIVE_ERROR"]] < 1e-11 and iparm[px.IPARM["STATIC_PIVOTING"]] == 0
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("facto", ["LLT", "LDLT", "LU"])
def test_pastix_irregular_graph_personal_ordering(facto):
    """A matrix that is not a grid: random sparse symmetric pattern (about 8 entries per row plus a few dense-ish
    rows), reverse Cuthill-McKee ordering handed over as API_ORDER_PERSONAL.  Exercises the symbolic stand-in and
    the plan on irregular supernodes (long skinny bloks, wide banded fronts)."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n = 4000
    rng = np.random.default_rng(77)
    i = rng.integers(0, n, 4 * n)
    j = np.clip(i + rng.integers(-60, 61, 4 * n), 0, n - 1)          # mostly local coupling ...
    far = rng.integers(0, n, n // 20)
    i = np.concatenate([i, far]); j = np.concatenate([j, rng.integers(0, n, n // 20)])   # ... and some long-range
    P = sp.coo_matrix((rng.standard_normal(len(i)), (i, j)), shape=(n, n)).tocsr()
    P = P + P.T
    A = (P + sp.diags(np.abs(P).sum(axis=1).A1 + 1.0)).tocsc()       # symmetric, diagonally dominant -> SPD
    if facto == "LU":
        A = (A + sp.triu(P, 1).multiply(0.3)).tocsc()                # unsymmetric values, symmetric pattern
    A.sort_indices()
    rcm = reverse_cuthill_mckee(sp.csr_matrix(A), symmetric_mode=True).astype(np.int64)   # new -> old
    invp = rcm + 1
    perm = np.empty(n, dtype=np.int64)
    perm[rcm] = np.arange(1, n + 1)
    M = A if facto == "LU" else sp.tril(A).tocsc()
    cp, r, v = M.indptr.astype(np.int64) + 1, M.indices.astype(np.int64) + 1, M.data.copy()
    x0 = rng.standard_normal(n)
    b = A @ x0
    rhs = b.copy()
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = getattr(px, "API_FACT_" + facto)
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if facto == "LU" else px.API_SYM_YES
    iparm[px.IPARM["ORDERING"]] = px.API_ORDER_PERSONAL
    iparm[px.IPARM["END_TASK"]] = px.API_TASK["SOLVE"]
    pd = px.pastix(None, n, cp, r, v, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0
    assert np.linalg.norm(A @ b - rhs) / np.linalg.norm(rhs) < 1e-11
    assert np.abs(b - x0).max() <= 1e-9 * np.abs(x0).max()
    iparm[px.IPARM["START_TASK"]] = iparm[px.IPARM["END_TASK"]] = px.API_TASK["CLEAN"]
    px.pastix(pd, n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("base", [1, 0])
@pytest.mark.parametrize("name,facto", [("orsirr_1030_lu", "LU"), ("rlap3d_12_llt", "LLT")])
def test_pastix_default_ordering_on_a_general_graph(name, facto, base, golden):
    """No API_ORDER_PERSONAL and no grid hint: the driver orders the graph itself (nested dissection by level
    structures, pastix_amd_order_graph -- the reference calls Scotch here), cuts the supernodes with blend's rule under
    IPARM_MIN/MAX_BLOCKSIZE, factorizes and solves.  orsirr.rua is the reference's own irregular fixture.  base 0: the
    same through a 0-based CSC (IPARM_BASEVAL = colptr[0])."""
    g = golden(name)
    n = int(g["n"])
    cp, r, v = g["colptr"].astype(np.int64), g["rows"].astype(np.int64), g["vals"].copy()
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    Afull = A if facto == "LU" else A + sp.tril(A, -1).T
    x0 = np.random.default_rng(4).standard_normal(n)
    b = Afull @ x0
    rhs = b.copy()
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = getattr(px, "API_FACT_" + facto)
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if facto == "LU" else px.API_SYM_YES
    iparm[px.IPARM["END_TASK"]] = px.API_TASK["REFINE"]
    cpb, rb = cp - (1 - base), r - (1 - base)
    pd = px.pastix(None, n, cpb, rb, v, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0
    assert iparm[px.IPARM["BASEVAL"]] == base
    assert np.array_equal(np.sort(perm), np.arange(n) + base) and np.array_equal(invp[perm - base] - base, np.arange(n))
    assert np.linalg.norm(Afull @ b - rhs) / np.linalg.norm(rhs) < 1e-10
    # the ordering is a fill-reducing one: far fewer factor entries than under the natural order
    nat = sy.symbolic(n, cp, r, None)["nnzl"]
    assert iparm[px.IPARM["NNZEROS"]] < (0.8 if name.startswith("orsirr") else 0.5) * nat
    iparm[px.IPARM["START_TASK"]] = iparm[px.IPARM["END_TASK"]] = px.API_TASK["CLEAN"]
    px.pastix(pd, n, cpb, rb, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("name", ["rlap3d_12_lu", "zrlap3d_12_ldlt", "zrlap3d_12_ldlh", "zrlap3d_8_lu"])
def test_device_refinement_all_modes_all_arithmetics(name, mode, golden):
    """pastix_amd_refine (csrc/refine.hip) directly: GMRES / CG / simple refinement / BiCGStab with vectors, SpMV and dot
    products on the device, on unsymmetric real, complex symmetric, complex Hermitian and complex unsymmetric matrices.
    The factors are perturbed by a large static-pivot threshold so that there is something to refine."""
    import ctypes
    from pastix_amd import COMPLEXDOUBLE, _lib
    g = golden(name)
    n = int(g["n"])
    cz = np.iscomplexobj(g["L0"])
    dt = np.complex128 if cz else np.float64
    cp, r = g["colptr"].astype(np.int64), g["rows"].astype(np.int64)
    v = np.ascontiguousarray(g["vals"], dtype=dt)
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    herm = g["facto"] == 3
    if g["sym"]:
        A = A + (sp.tril(A, -1).conj().T if herm else sp.tril(A, -1).T)
    rng = np.random.default_rng(9)
    x0 = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cz else 0)).astype(dt)
    b = np.ascontiguousarray(A @ x0, dtype=dt)
    perm = g["perm"].astype(np.int64)
    c4 = g["cblk4"]
    wid = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(wid * c4[:-1, 3])])
    piv = np.concatenate([np.abs(g["L1"][off[k] + np.arange(wid[k]) * (c4[k, 3] + 1)]) for k in range(len(wid))])
    crit = float(np.sort(piv)[n // 8])             # an eighth of the reference's pivots lie below: they get clamped
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE if cz else 1) as p:
        p.fill_csc(g["sym"], n, cp, r, v, perm)
        st = p.factorize(crit)
        assert st["nbpivot"] > 0                                   # perturbed factors
        bp = np.empty(n, dtype=dt)
        bp[perm] = b
        x = np.ascontiguousarray(p.solve(bp)[perm], dtype=dt)
        first = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
        it = ctypes.c_int64(0)
        err = ctypes.c_double(0)
        rc = _lib.lib().pastix_amd_refine(p._h, mode, 2 if (g["sym"] and herm) else int(g["sym"]), ctypes.c_int64(n),
                                          _lib.ptr(cp), _lib.ptr(r), _lib.ptr(v), _lib.ptr(perm), _lib.ptr(b), _lib.ptr(x),
                                          ctypes.c_int64(1), ctypes.c_double(1e-12), ctypes.c_int64(250), 25,
                                          ctypes.byref(it), ctypes.byref(err))
        assert rc == 0
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    assert first > 1e-8 and res < 1e-11 and err.value < 1e-11 and 0 < it.value <= 250


@pytest.mark.parametrize("facto", ["LLT", "LU"])
def test_pastix_fill_matrix_mode(facto):
    """iparm[IPARM_FILL_MATRIX] = API_YES through pastix(): a structure-only run of the reference (pastix.c:3282,
    coefinit.c:343-443, critere sopalin3d.c:597-598) -- the values of the CSC must not be read (they are NaN here), the
    run succeeds without static pivots and reports its time.  (Factor values of this mode: test_fake_fill_matches_reference.)"""
    N = 8
    n, cp, r, v = sy.laplacian_3d(N, full=(facto == "LU"))
    perm0, invp0 = sy.order_grid(N, N, N)
    perm, invp = perm0 + 1, invp0 + 1
    b = np.ones(n)
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FACTORIZATION"]] = getattr(px, "API_FACT_" + facto)
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if facto == "LU" else px.API_SYM_YES
    iparm[px.IPARM["ORDERING"]] = px.API_ORDER_PERSONAL
    iparm[px.IPARM["FILL_MATRIX"]] = px.API_YES
    iparm[px.IPARM["START_TASK"]] = px.API_TASK["ORDERING"]
    iparm[px.IPARM["END_TASK"]] = px.API_TASK["NUMFACT"]
    pd = px.pastix(None, n, cp, r, np.full_like(v, np.nan), perm, invp, b, 1, iparm, dparm)   # values must not be read
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0
    assert iparm[px.IPARM["STATIC_PIVOTING"]] == 0 and dparm[px.DPARM["FACT_TIME"]] > 0
    iparm[px.IPARM["START_TASK"]] = iparm[px.IPARM["END_TASK"]] = px.API_TASK["CLEAN"]
    px.pastix(pd, n, cp, r, v, perm, invp, b, 1, iparm, dparm)


@pytest.mark.parametrize("facto", ["LLT", "LDLT", "LU"])
def test_s_pastix_entry_point(facto):
    """S_pastix (iparm[IPARM_FLOAT] = API_REALSINGLE, api.h:522-525): float values and right-hand side through the entry
    point; the factorization runs on the native fp32 engine, the refinement (double vectors inside) brings the residual
    of the FLOAT system down to float rounding."""
    import scipy.sparse as sp
    N = 14
    n, cp, r, v = sy.laplacian_3d(N)
    perm0, invp0 = sy.order_grid(N, N, N)
    perm, invp = perm0 + 1, invp0 + 1
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    if facto == "LU":
        Af = (A + sp.tril(A, -1).T + sp.triu(A + sp.tril(A, -1).T, 1).multiply(0.1)).tocsc()
        Af.sort_indices()
        cp, r, v = Af.indptr.astype(np.int64) + 1, Af.indices.astype(np.int64) + 1, Af.data
    else:
        Af = (A + sp.tril(A, -1).T).tocsc()
    v32 = v.astype(np.float32)
    b = np.random.default_rng(9).random(n).astype(np.float32)
    rhs = b.copy()
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FLOAT"]] = px.API_REALSINGLE
    iparm[px.IPARM["FACTORIZATION"]] = getattr(px, "API_FACT_" + facto)
    iparm[px.IPARM["SYM"]] = px.API_SYM_NO if facto == "LU" else px.API_SYM_YES
    iparm[px.IPARM["ORDERING"]] = px.API_ORDER_PERSONAL
    iparm[px.IPARM["REFINEMENT"]] = px.API_RAF_GMRES
    dparm[px.DPARM["EPSILON_REFINEMENT"]] = 1e-10
    pd = _run_tasks(None, "ORDERING", "REFINE", n, cp, r, v32, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0 and b.dtype == np.float32
    A32 = sp.csc_matrix((v32.astype(np.float64), Af.indices if facto == "LU" else A.indices, Af.indptr if facto == "LU" else A.indptr), shape=(n, n))
    if facto != "LU":
        A32 = A32 + sp.tril(A32, -1).T
    res = np.linalg.norm(A32 @ b.astype(np.float64) - rhs) / np.linalg.norm(rhs)
    assert res < 5e-6, res                                    # (x is rounded to float: ~1e-7 relative)
    assert dparm[px.DPARM["RELATIVE_ERROR"]] < 1e-9            # the refinement itself converged in double
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v32, perm, invp, b, 1, iparm, dparm)


def test_c_pastix_entry_point():
    """C_pastix (API_COMPLEXSINGLE): float complex values; no native complex-single engine -- the fp64 engine computes on
    the widened values, the solution is rounded to complex64."""
    import scipy.sparse as sp
    N = 5
    n, cp, r, v, _ = sy.elasticity_3d(N)
    v64 = v.astype(np.complex64)
    A = sp.csc_matrix((v64.astype(np.complex128), r - 1, cp - 1), shape=(n, n))
    Afull = (A + sp.tril(A, -1).T).tocsc()
    rng = np.random.default_rng(3)
    b = (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
    rhs = b.copy()
    iparm, dparm = px.init_param()
    iparm[px.IPARM["FLOAT"]] = px.API_COMPLEXSINGLE
    iparm[px.IPARM["FACTORIZATION"]] = px.API_FACT_LDLT
    iparm[px.IPARM["SYM"]] = px.API_SYM_YES
    perm = np.zeros(n, dtype=np.int64)
    invp = np.zeros(n, dtype=np.int64)
    pd = _run_tasks(None, "ORDERING", "SOLVE", n, cp, r, v64, perm, invp, b, 1, iparm, dparm)
    assert iparm[px.IPARM["ERROR_NUMBER"]] == 0 and b.dtype == np.complex64
    assert np.linalg.norm(Afull @ b.astype(np.complex128) - rhs) / np.linalg.norm(rhs) < 5e-6
    _run_tasks(pd, "CLEAN", "CLEAN", n, cp, r, v64, perm, invp, b, 1, iparm, dparm)

