"""Pin the oracle (oracle/sopalin_oracle.c) against outputs of the REAL reference
(tests/golden/*.npz, made by tests/golden/make_golden.py from oracle/_ref)."""
import numpy as np
import pytest

import oracle_lib
from conftest import golden_names

TOL = 1e-12  # per entry, relative to max|L_ref| (SURVEY 8d parity tolerance for d)


@pytest.mark.parametrize("name", golden_names(prec=None))
def test_fill_matches_reference(name, golden):
    g = golden(name)
    if name.startswith("fake_"):        # IPARM_FILL_MATRIX = API_YES: no CSC values (coefinit.c:343-443)
        L0, U0 = oracle_lib.fill_fake(g["facto"], g["n"], g["cblk4"])
        assert g["critere"] == (g["n"] ** 2 + g["n"]) * np.sqrt(1e-31)       # sopalin3d.c:597-598
    else:
        L0, U0 = oracle_lib.fill(g["facto"], g["sym"], g["n"], g["colptr"], g["rows"], g["vals"],
                                 g["perm"], g["cblk4"], g["blok4"])
    assert np.array_equal(L0, g["L0"])          # a scatter of input values: exact
    if g["facto"] == 2:
        assert np.array_equal(U0, g["U0"])


@pytest.mark.parametrize("name", golden_names(prec=None))
def test_factor_matches_reference(name, golden):
    g = golden(name)
    L1, U1, nbpiv = oracle_lib.sopalin(g["facto"], g["cblk4"], g["blok4"], g["L0"], g["U0"], g["critere"])
    scale = np.abs(g["L1"]).max()
    assert np.abs(L1 - g["L1"]).max() <= TOL * scale
    if g["facto"] == 2:
        assert np.abs(U1 - g["U1"]).max() <= TOL * max(scale, np.abs(g["U1"]).max())
    assert nbpiv == g["nbpivot"]


@pytest.mark.parametrize("name", golden_names(prec=None))
def test_flops_match_reference(name, golden):
    g = golden(name)
    f = oracle_lib.fact_flops(g["facto"], g["prec"], g["cblk4"], g["blok4"])
    assert abs(f - g["flops"]) <= 1e-9 * g["flops"]


@pytest.mark.parametrize("name", golden_names(prec=None))
def test_solve_matches_reference(name, golden):
    g = golden(name)
    bp = np.empty_like(g["b"])
    bp[g["perm"]] = g["b"]
    xp = oracle_lib.solve(g["facto"], g["cblk4"], g["blok4"], g["L1"], g["U1"], bp)
    x = xp[g["perm"]]
    assert np.abs(x - g["x"]).max() <= 1e-9 * np.abs(g["x"]).max()


def test_static_pivot_count():
    """Tiny 2x2 with a zero pivot: the clamp path (compute_diag.c:133-137) fires once."""
    c4 = np.array([[0, 1, 0, 2], [2, 2, 1, 0]], dtype=np.int64)
    b4 = np.array([[0, 1, 0, 0]], dtype=np.int64)
    L = np.array([4.0, 2.0, 0.0, 1.0])   # A = [[4,2],[2,1]] -> second pivot exactly 0
    L1, _, nb = oracle_lib.sopalin(0, c4, b4, L, None, 1e-8)
    assert nb == 1
    assert L1[0] == 2.0 and L1[1] == 1.0 and abs(L1[3] - 1e-4) < 1e-18
