"""Host-side layout producer (no GPU): the emitted cblk/blok tables are a valid SolverMatrix layout
and factorizing on them with the CPU oracle solves the system."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_lib
from pastix_amd import _lib, fact_flops
from pastix_amd import symbolic as sy


def _residual(n, cp, r, v, x, b):
    A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
    A = A + sp.tril(A, -1).T
    return np.linalg.norm(A @ x - b) / np.linalg.norm(b)


@pytest.mark.parametrize("dims,leaf,bs,am", [((6, 6, 6), 8, 128, 5), ((9, 7, 5), 8, 16, 5), ((12, 12, 12), 27, 32, 10),
                                             ((16, 16, 16), 8, 128, 0), ((40, 1, 1), 8, 8, 5)])
def test_layout_valid_and_solves(dims, leaf, bs, am):
    n, cp, r, v = sy.laplacian_3d(*dims)
    perm, invp = sy.order_grid(*dims, leaf=leaf)
    assert np.array_equal(perm[invp], np.arange(n))
    s = sy.symbolic(n, cp, r, perm, max_blocksize=bs, amalgamation_pct=am)
    c4, b4 = s["cblk4"], s["blok4"]
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    assert w.max() <= bs and c4[-1, 2] == len(b4)
    assert np.array_equal(s["perm"][s["invp"]], np.arange(n))
    # the engine's own validator accepts it (plan_create stops at "no device" only after validation)
    la = _lib.LayoutArrays(c4, b4)
    h = ctypes.c_void_p()
    rc = _lib.lib().pastix_amd_plan_create(ctypes.byref(la.c), 0, 1, None, ctypes.byref(h))
    assert rc in (0, -3)
    if rc == 0:
        _lib.lib().pastix_amd_plan_destroy(h)
    L0, _ = oracle_lib.fill(0, 1, n, cp, r, v, s["perm"], c4, b4)
    L1, _, nb = oracle_lib.sopalin(0, c4, b4, L0, None, 1e-14)
    assert nb == 0
    b = np.random.default_rng(1).random(n)
    bp = np.empty(n)
    bp[s["perm"]] = b
    x = oracle_lib.solve(0, c4, b4, L1, None, bp)[s["perm"]]
    assert _residual(n, cp, r, v, x, b) < 1e-12
    # nnz(L) reported = entries of the lower trapezoids
    assert s["nnzl"] == int((c4[:-1, 3] * w - w * (w - 1) // 2).sum())
    assert fact_flops(c4, b4, 0) > 0


def test_amalgamation_reduces_supernodes():
    n, cp, r, v = sy.laplacian_3d(14)
    perm, _ = sy.order_grid(14, 14, 14)
    s0 = sy.symbolic(n, cp, r, perm, amalgamation_pct=0)
    s5 = sy.symbolic(n, cp, r, perm, amalgamation_pct=5)
    assert s5["nsuper_amalg"] < s0["nsuper_amalg"] <= s0["nsuper_fund"]
    assert s0["nnzl"] <= s5["nnzl"]      # pct=0 only performs zero-fill merges
    assert s5["nnzl"] <= 1.06 * s0["nnzl"]


def test_bad_input_rejected():
    n, cp, r, v = sy.laplacian_3d(4)
    bad = np.zeros(n, dtype=np.int64)
    with pytest.raises(_lib.PastixAmdError):
        sy.symbolic(n, cp, r, bad)


def test_schur_unknowns_stay_one_last_cblk():
    """symbolic(schur_n): the last schur_n unknowns (coupled pairwise in the pattern) end up as ONE cblk at the end,
    whatever its width, and no other column joins it."""
    N = 9
    n, cp, r, v = sy.laplacian_3d(N)
    perm0, _ = sy.order_grid(N, N, N)
    for ns in (1, 7, 300):
        schur = np.sort(np.random.default_rng(3).choice(n, size=ns, replace=False))
        is_s = np.zeros(n, bool)
        is_s[schur] = True
        order = np.argsort(perm0)                                  # new -> old
        perm = np.empty(n, dtype=np.int64)
        perm[order[~is_s[order]]] = np.arange(n - ns)
        perm[order[is_s[order]]] = np.arange(n - ns, n)
        cols, cp2 = [], [1]
        for j in range(n):
            cols += list(r[cp[j] - 1:cp[j + 1] - 1])
            if is_s[j]:
                cols += [u + 1 for u in schur if u > j]
            cp2.append(len(cols) + 1)
        s = sy.symbolic(n, np.array(cp2), np.array(cols), perm, max_blocksize=64, schur_n=ns)
        c4 = s["cblk4"]
        assert c4[-2][0] == n - ns and c4[-2][1] == n - 1          # last real cblk = the Schur block, unsplit
        assert sorted(np.argsort(s["perm"])[n - ns:].tolist()) == schur.tolist()
        assert ((c4[:-2, 1] - c4[:-2, 0] + 1) <= 64).all()
