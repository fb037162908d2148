"""Edge cases of the HIP path (hand-made layouts): smallest problems, single dense cblks of every width class,
one-column cblks, error codes.  The CPU oracle (pinned against the reference, test_oracle_golden.py) is the checker."""
import numpy as np
import pytest

import oracle_lib
from pastix_amd import Plan
from pastix_amd._lib import PastixAmdError

pytestmark = pytest.mark.gpu
TOL = 1e-12


def dense_layout(widths):
    """Dense lower-triangular block structure: cblk k has a diagonal blok and one blok facing every later cblk."""
    first = np.concatenate([[0], np.cumsum(widths)]).astype(np.int64)
    n = int(first[-1])
    nc = len(widths)
    cblk, blok = [], []
    for k in range(nc):
        off = 0
        cblk.append([first[k], first[k + 1] - 1, len(blok), n - first[k]])
        for t in range(k, nc):
            blok.append([first[t], first[t + 1] - 1, t, off])
            off += widths[t]
    cblk.append([n, n, len(blok), 0])
    return np.array(cblk, dtype=np.int64), np.array(blok, dtype=np.int64), n


def panels_of(A, c4):
    """Pack the lower part of dense A (rows >= first column of the cblk) into the panel arena."""
    out = []
    for k in range(len(c4) - 1):
        f, l = int(c4[k, 0]), int(c4[k, 1])
        out.append(A[f:, f:l + 1].flatten(order="F"))
    return np.concatenate(out)


def lower_mask(c4):
    m = []
    for k in range(len(c4) - 1):
        w, s = int(c4[k, 1] - c4[k, 0] + 1), int(c4[k, 3])
        blk = np.ones((s, w), dtype=bool)
        blk[:w, :w] = np.tril(np.ones((w, w), dtype=bool))
        m.append(blk.flatten(order="F"))
    return np.concatenate(m)


def spd(n, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, n))
    return B @ B.T + n * np.eye(n)


@pytest.mark.parametrize("facto", [0, 1, 2])
def test_one_by_one(facto):
    c4, b4, n = dense_layout([1])
    L0 = np.array([4.0])
    U0 = np.array([4.0]) if facto == 2 else None
    with Plan(c4, b4, facto) as p:
        p.upload(L0, U0)
        st = p.factorize(1e-30)
        L1, U1 = p.download()
    assert st["nbpivot"] == 0
    assert L1[0] == (2.0 if facto == 0 else 4.0)
    if facto == 2:
        assert U1[0] == 4.0


@pytest.mark.parametrize("widths", [[7], [16], [17], [64], [100], [128], [129], [200], [256],
                                     [1, 1, 1], [5, 1, 130, 1], [128, 128, 3], [256, 40]])
@pytest.mark.parametrize("facto", [0, 1, 2])
def test_dense_blocks_match_oracle(widths, facto):
    c4, b4, n = dense_layout(widths)
    A = spd(n, 11 + n)
    if facto == 2:
        A = A + np.triu(np.random.default_rng(5).standard_normal((n, n)), 1) * 0.1     # unsymmetric values
    L0 = panels_of(A, c4)
    U0 = panels_of(A.T, c4) if facto == 2 else None
    Lo, Uo, nbo = oracle_lib.sopalin(facto, c4, b4, L0, U0, 1e-30)
    with Plan(c4, b4, facto) as p:
        p.upload(L0, U0)
        st = p.factorize(1e-30)
        L1, U1 = p.download()
    assert st["nbpivot"] == nbo == 0
    m = lower_mask(c4) if facto != 2 else np.ones(L1.size, bool)
    assert np.abs(L1 - Lo)[m].max() <= TOL * np.abs(Lo[m]).max()
    if facto == 2:
        assert np.abs(U1 - Uo).max() <= TOL * np.abs(Uo).max()
    if facto == 0 and len(widths) == 1:      # independent check: dense Cholesky
        C = np.linalg.cholesky(A)
        assert np.abs(L1.reshape(n, n, order="F") - C)[np.tril_indices(n)].max() <= 1e-11 * np.abs(C).max()


@pytest.mark.parametrize("widths", [[257], [334, 20], [300, 177, 40], [40, 600, 30, 290]])
@pytest.mark.parametrize("facto", [0, 1, 2])
def test_cblks_wider_than_256_are_factorized_in_column_groups(widths, facto):
    """The reference's blend can leave supernodes wider than the panel kernels' 256 columns (334 on its own
    orsirr.rua fixture): the engine cuts them into column groups internally; panels go in and come out in the
    caller's layout."""
    c4, b4, n = dense_layout(widths)
    A = spd(n, 31 + n)
    if facto == 2:
        A = A + np.triu(np.random.default_rng(9).standard_normal((n, n)), 1) * 0.1
    L0 = panels_of(A, c4)
    U0 = None
    if facto == 2:                                  # the reference's fill: the square A_kk in coeftab, zeros in ucoeftab
        U0 = panels_of(A.T, c4)
        o = 0
        for k in range(len(c4) - 1):
            wk, sk = int(c4[k, 1] - c4[k, 0] + 1), int(c4[k, 3])
            U0[o:o + wk * sk].reshape(wk, sk)[:, :wk] = 0.0
            o += wk * sk
    Lo, Uo, nbo = oracle_lib.sopalin(facto, c4, b4, L0, U0, 1e-30)
    with Plan(c4, b4, facto) as p:
        p.upload(L0, U0)
        La, Ua = p.download()                       # round trip through the column groups before factorizing
        st = p.factorize(1e-30)
        L1, U1 = p.download()
    lm = lower_mask(c4)
    if facto == 2:
        assert np.array_equal(La, L0) and np.array_equal(Ua, U0)
    else:
        assert np.array_equal(La[lm], L0[lm])
    assert st["nbpivot"] == nbo == 0
    m = lm if facto != 2 else np.ones(L1.size, bool)
    assert np.abs(L1 - Lo)[m].max() <= TOL * np.abs(Lo[m]).max()
    if facto == 2:
        assert np.abs(U1 - Uo).max() <= TOL * np.abs(Uo).max()      # incl. ucoeftab's diagonal blok = coeftab's transposed
    # and the solve works on the column groups
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    b = A @ x
    with Plan(c4, b4, facto) as p:
        p.upload(L0, U0)
        p.factorize(1e-30)
        xs = p.solve(b.copy())
    assert np.abs(xs - x).max() <= 1e-9 * np.abs(x).max()


@pytest.mark.parametrize("widths", [[334, 20], [300, 177, 40], [40, 600, 30, 290]])
@pytest.mark.parametrize("facto", [0, 1, 2])
def test_wide_cblks_through_the_one_shot_entry_points(widths, facto):
    """The same layouts through {po,sy,ge}_sopalin with one host buffer per cblk: the staging cuts a re-cut cblk's panel into
    its column groups' panels on the way in and joins them on the way out (LU: the diagonal blok's upper blocks come back as
    the transposes held by the other arena)."""
    from pastix_amd.solver import sopalin_tabs
    from pastix_amd import _lib
    c4, b4, n = dense_layout(widths)
    A = spd(n, 77 + n)
    if facto == 2:
        A = A + np.triu(np.random.default_rng(5).standard_normal((n, n)), 1) * 0.1
    L0 = panels_of(A, c4)
    U0 = None
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    if facto == 2:
        U0 = panels_of(A.T, c4)
        for k in range(len(w)):
            U0[off[k]:off[k + 1]].reshape(int(w[k]), int(c4[k, 3]))[:, :int(w[k])] = 0.0
    Lo, Uo, nbo = oracle_lib.sopalin(facto, c4, b4, L0, U0, 1e-30)
    tabs = [L0[off[k]:off[k + 1]].copy() for k in range(len(w))]
    utabs = [U0[off[k]:off[k + 1]].copy() for k in range(len(w))] if facto == 2 else None
    try:
        st = sopalin_tabs(facto, c4, b4, tabs, utabs, critere=1e-30)
    finally:
        _lib.lib().pastix_amd_release_cached_plan()
    L1 = np.concatenate(tabs)
    assert st["nbpivot"] == nbo == 0
    lm = lower_mask(c4)
    m = lm if facto != 2 else np.ones(L1.size, bool)
    assert np.abs(L1 - Lo)[m].max() <= TOL * np.abs(Lo[m]).max()
    if facto == 2:
        assert np.abs(np.concatenate(utabs) - Uo).max() <= TOL * np.abs(Uo).max()


@pytest.mark.parametrize("facto", [1, 3, 2])
def test_complex_cblks_wider_than_256(facto):
    """The column-group path on the split planes: z LDLt (complex symmetric), LDLh (Hermitian) and LU."""
    from pastix_amd import COMPLEXDOUBLE
    widths = [300, 20, 290]
    c4, b4, n = dense_layout(widths)
    rng = np.random.default_rng(17)
    B = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    if facto == 3:
        A = B @ B.conj().T + 4 * n * np.eye(n)             # Hermitian positive definite
    else:
        A = B @ B.T * 0.05 + 4 * n * np.eye(n)             # complex symmetric, diagonally dominant
    if facto == 2:
        A = A + np.triu(rng.standard_normal((n, n)), 1) * 0.1
    L0 = panels_of(A, c4).astype(np.complex128)
    U0 = None
    if facto == 2:
        U0 = panels_of(A.T, c4).astype(np.complex128)
        o = 0
        for k in range(len(c4) - 1):
            wk, sk = int(c4[k, 1] - c4[k, 0] + 1), int(c4[k, 3])
            U0[o:o + wk * sk].reshape(wk, sk)[:, :wk] = 0.0
            o += wk * sk
    Lo, Uo, nbo = oracle_lib.sopalin(facto, c4, b4, L0, U0, 1e-30)
    with Plan(c4, b4, facto, floattype=COMPLEXDOUBLE) as p:
        p.upload(L0, U0)
        st = p.factorize(1e-30)
        L1, U1 = p.download()
    assert st["nbpivot"] == nbo == 0
    m = lower_mask(c4) if facto != 2 else np.ones(L1.size, bool)
    assert np.abs(L1 - Lo)[m].max() <= TOL * np.abs(Lo[m]).max()
    if facto == 2:
        assert np.abs(U1 - Uo).max() <= TOL * np.abs(Uo).max()


def test_bad_layout_is_rejected():
    c4, b4, n = dense_layout([8, 8])
    b4 = b4.copy()
    b4[1, 2] = 0                       # off-diagonal blok facing its own cblk
    with pytest.raises(PastixAmdError) as e:
        Plan(c4, b4, 0)
    assert e.value.code in (-6, -1)


def test_indefinite_llt_reports_numeric_error():
    """The reference's check after sqrt is ineffective (compute_diag.c:143, NaNs propagate); here a non-finite
    pivot is reported as PASTIX_AMD_ERR_NUMERIC."""
    c4, b4, n = dense_layout([40, 24])
    A = spd(n, 3)
    A[10, 10] = -50.0
    with Plan(c4, b4, 0) as p:
        p.upload(panels_of(A, c4))
        st = p.factorize(1e-30, allow_numeric_error=True)
    assert st["rc"] == -4


def test_all_pivots_clamped():
    """Zero matrix: every pivot is below critere and is replaced by it (compute_diag.c:133-137)."""
    c4, b4, n = dense_layout([20, 12])
    L0 = np.zeros(int(((c4[:-1, 1] - c4[:-1, 0] + 1) * c4[:-1, 3]).sum()))
    Lo, _, nbo = oracle_lib.sopalin(0, c4, b4, L0, None, 0.5)
    with Plan(c4, b4, 0) as p:
        p.upload(L0)
        st = p.factorize(0.5)
        L1, _ = p.download()
    assert st["nbpivot"] == nbo == n
    assert np.abs(L1 - Lo)[lower_mask(c4)].max() <= TOL


@pytest.mark.parametrize("facto", [0, 1, 2])
@pytest.mark.parametrize("widths", [[40, 30, 300], [128, 128, 64], [17, 500]])
def test_schur_mode_leaves_the_schur_complement_in_the_last_cblk(widths, facto):
    """IPARM_SCHUR at the sopalin boundary (compute_1d, sopalin_compute.c:767-772): the last cblk is updated but not
    factorized, so its diagonal blok is S = A22 - A21 A11^-1 A12 (what pastix_getSchur returns); it may be wider
    than 256 columns.  The other cblks are factorized as usual."""
    c4, b4, n = dense_layout(widths)
    A = spd(n, 21 + n)
    if facto == 2:
        A = A + np.triu(np.random.default_rng(7).standard_normal((n, n)), 1) * 0.1
    L0 = panels_of(A, c4)
    U0 = panels_of(A.T, c4) if facto == 2 else None
    with Plan(c4, b4, facto, schur=True) as p:
        p.upload(L0, U0)
        st = p.factorize(1e-30)
        L1, U1 = p.download()
    assert st["nbpivot"] == 0
    ns = widths[-1]
    n1 = n - ns
    S = A[n1:, n1:] - A[n1:, :n1] @ np.linalg.solve(A[:n1, :n1], A[:n1, n1:])
    got = L1[-ns * ns:].reshape(ns, ns, order="F")
    if facto == 2:
        # LU: contributions to a diagonal blok from the U side are added transposed into coeftab
        # (add_contrib_local, sopalin_compute.c:430-435,572-575), so the whole square of S sits in the L arena
        assert np.abs(got - S).max() <= 1e-10 * np.abs(S).max()
    else:
        tri = np.tril_indices(ns)
        assert np.abs(got - S)[tri].max() <= 1e-10 * np.abs(S).max()
    # the factorized part equals an ordinary factorization of the same layout
    Lo, Uo, _ = oracle_lib.sopalin(facto, c4, b4, L0, U0, 1e-30)
    m = lower_mask(c4) if facto != 2 else np.ones(L1.size, bool)
    m[-ns * ns:] = False
    assert np.abs(L1 - Lo)[m].max() <= TOL * np.abs(Lo[m]).max()
