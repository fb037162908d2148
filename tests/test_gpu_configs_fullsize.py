"""BASELINE.json's configurations at (or next to) their stated sizes, through the C ABI on the MI355X: the parity
suite pins the factors at fixture sizes against the reference's own output; here the size-independent properties the
domain offers are checked where the oracle cannot follow -- ||Ax - b|| / ||b|| <= 1e-10 after one device solve (SURVEY
8d), no static pivot, and a bitwise-equal refactorization (tile ownership: the summation order is part of the plan).
configs[1] = 100^3 dLLt; configs[2] = dLU with static pivoting (80^3 here: the test suite's memory and time budget; the
bench line carries 192^3, the largest grid whose L and U panels fit one device); configs[4] = z LDLt on the 3-dof
elasticity pattern (40^3 nodes, n = 192 000)."""
import numpy as np
import pytest
import scipy.sparse as sp

from pastix_amd import COMPLEXDOUBLE, Plan
from pastix_amd import symbolic as sy

pytestmark = pytest.mark.gpu


def _run(n, cp, r, v, perm, facto, ftype, crit, full):
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    c4, b4 = s["cblk4"], s["blok4"]
    cz = ftype == COMPLEXDOUBLE
    with Plan(c4, b4, facto, floattype=ftype) as p:
        p.fill_csc(0 if full else 1, n, cp, r, v, s["perm"])
        st = p.factorize(crit)
        La, Ua = p.download()
        rng = np.random.default_rng(7)
        b = rng.random(n) + (1j * rng.random(n) if cz else 0)
        bp = np.empty(n, dtype=b.dtype)
        bp[s["perm"]] = b
        x = p.solve(bp)[s["perm"]]
        p.refill()
        st2 = p.factorize(crit)
        Lb, Ub = p.download()
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    Ax = A @ x if full else A @ x + sp.tril(A, -1).T @ x
    resid = float(np.linalg.norm(Ax - b) / np.linalg.norm(b))
    assert st["nbpivot"] == 0 and st2["nbpivot"] == 0
    assert resid <= 1e-10, resid
    assert np.array_equal(La, Lb)
    if Ua is not None:
        assert np.array_equal(Ua, Ub)
    return st


def test_config2_laplacian_100_dllt():
    N = 100
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    st = _run(n, cp, r, v, perm, 0, 1, 6.0 * 2 * np.sqrt(1e-31), False)
    assert st["fact_time"] < 1.0              # (0.14 s on an MI355X: a fallback to anything else would show)


def test_config3_laplacian_80_dlu_static_pivoting():
    N = 80
    n, cp, r, v = sy.laplacian_3d(N, full=True)
    perm, _ = sy.order_grid(N, N, N)
    _run(n, cp, r, v, perm, 2, 1, 6.0 * 2 * np.sqrt(1e-31), True)


def test_config5_elasticity_40_zldlt():
    N = 40
    n, cp, r, v, _ = sy.elasticity_3d(N)
    perm, _ = sy.order_grid_dof(N, 3)
    _run(n, cp, r, v, perm, 1, COMPLEXDOUBLE, 1e-12, False)
