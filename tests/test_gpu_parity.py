"""GPU parity: the HIP path (through the C ABI) against the golden outputs of the real reference
and against the CPU oracle on the same inputs."""
import numpy as np
import pytest

import oracle_lib
from conftest import golden_names, recut_mask
from pastix_amd import COMPLEXDOUBLE, Plan, sopalin_tabs

pytestmark = pytest.mark.gpu

TOL = 1e-12   # per entry relative to max|L_ref|


@pytest.mark.parametrize("lookahead", [1, 300, 1 << 30])
@pytest.mark.parametrize("name", golden_names("llt"))
def test_llt_matches_reference_golden(name, lookahead, golden):
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], lookahead=lookahead) as p:
        p.upload(g["L0"])
        st = p.factorize(g["critere"])
        L1, _ = p.download()
    scale = np.abs(g["L1"]).max()
    assert np.abs(L1 - g["L1"])[recut_mask(g["cblk4"])].max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", golden_names("llt"))
def test_device_fill_matches_reference(name, golden):
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"]) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        L0, _ = p.download()
    assert np.array_equal(L0, g["L0"])


def test_one_shot_tabs_dropin(golden):
    g = golden("rlap3d_10_llt")
    c4 = g["cblk4"]
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    off = np.concatenate([[0], np.cumsum(w * c4[:-1, 3])])
    tabs = [g["L0"][off[k]:off[k + 1]].copy() for k in range(len(w))]
    st = sopalin_tabs(0, c4, g["blok4"], tabs, critere=g["critere"])
    L1 = np.concatenate(tabs)
    assert np.abs(L1 - g["L1"])[recut_mask(c4)].max() <= TOL * np.abs(g["L1"]).max()
    assert st["fact_time"] > 0


def test_static_pivot_clamp_matches_oracle(golden):
    """Force the clamp path (compute_diag.c:133-137): with critere above the natural pivots of the
    Laplacian every small pivot is raised to critere (the matrix stays positive definite); the clamp
    count and the factors must match the oracle."""
    g = golden("lap3d_8_llt")
    c4 = g["cblk4"]
    crit = 5.9
    Lo, _, nbo = oracle_lib.sopalin(0, c4, g["blok4"], g["L0"], None, crit)
    with Plan(c4, g["blok4"], 0) as p:
        p.upload(g["L0"])
        st = p.factorize(crit)
        L1, _ = p.download()
    assert nbo >= 10 and st["nbpivot"] == nbo
    assert np.isfinite(Lo).all()
    assert np.abs(L1 - Lo).max() <= TOL * np.abs(Lo).max()


@pytest.mark.parametrize("name,crit", [("rlap3d_12_ldlt", 5.9), ("rlap3d_20_lu_bs128", 5.9), ("rlap3d_20_llt_bs128", 5.9),
                                       ("zrlap3d_20_ldlt_bs128", 5.9), ("zrlap3d_12_ldlh", 5.9)])
def test_static_pivot_clamp_of_every_diagonal_kernel_matches_oracle(name, crit, golden):
    """The clamp is a rarely taken branch of the round-4 diagonal kernels (pivot read from the accumulator layout, the
    clamped value written back into the tile before the column is scaled): LDLt, LU, LLt on 128-wide cblks and complex
    LDLt / LDLh with critere above the natural pivots -- clamp counts and factors equal the oracle's."""
    g = golden(name)
    c4, b4, facto = g["cblk4"], g["blok4"], int(g["facto"])
    cz = np.iscomplexobj(g["L0"])
    U0 = g["U0"] if facto == 2 else None
    Lo, Uo, nbo = oracle_lib.sopalin(facto, c4, b4, g["L0"], U0, crit)
    kw = {"floattype": COMPLEXDOUBLE} if cz else {}
    with Plan(c4, b4, facto, **kw) as p:
        p.upload(g["L0"], U0) if facto == 2 else p.upload(g["L0"])
        st = p.factorize(crit)
        L1, U1 = p.download()
    assert nbo >= 10 and st["nbpivot"] == nbo
    m = _lower_mask(c4) if facto in (1, 3) else recut_mask(c4) if facto == 0 else np.ones(len(Lo), bool)
    assert np.isfinite(Lo[m]).all()
    assert np.abs(L1 - Lo)[m].max() <= TOL * np.abs(Lo[m]).max()
    if facto == 2:
        assert np.abs(U1 - Uo).max() <= TOL * max(np.abs(Uo).max(), np.abs(Lo).max())


def _lower_mask(c4):
    """True where a panel entry is meaningful for LDLt: everything except the strict upper triangle
    of the diagonal bloks (the reference's blocked sytrf leaves GEMM by-products there)."""
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    parts = []
    for k in range(len(w)):
        s, wk = int(c4[k, 3]), int(w[k])
        m = np.ones((wk, s), dtype=bool)           # [col][row]
        for c in range(wk):
            m[c, :c] = False
        parts.append(m.ravel())
    return np.concatenate(parts)


@pytest.mark.parametrize("lookahead", [1, 0])
@pytest.mark.parametrize("name", golden_names("ldlt"))
def test_ldlt_matches_reference_golden(name, lookahead, golden):
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], lookahead=lookahead) as p:
        p.upload(g["L0"])
        st = p.factorize(g["critere"])
        L1, _ = p.download()
    m = _lower_mask(g["cblk4"])
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"]
    assert st["inertia"] == g["inertia"] == g["n"]       # SPD input: all D positive (IPARM_INERTIA)


@pytest.mark.parametrize("lookahead", [1, 0])
@pytest.mark.parametrize("name", golden_names("lu"))
def test_lu_matches_reference_golden(name, lookahead, golden):
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], lookahead=lookahead) as p:
        p.upload(g["L0"], g["U0"])
        st = p.factorize(g["critere"])
        L1, U1 = p.download()
    scale = max(np.abs(g["L1"]).max(), np.abs(g["U1"]).max())
    assert np.abs(L1 - g["L1"]).max() <= TOL * scale
    assert np.abs(U1 - g["U1"]).max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", golden_names("lu"))
def test_lu_device_fill_matches_reference(name, golden):
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"]) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        L0, U0 = p.download()
    assert np.array_equal(L0, g["L0"]) and np.array_equal(U0, g["U0"])


@pytest.mark.parametrize("name", golden_names("ldlt", prec="z"))
def test_z_ldlt_matches_reference_golden(name, golden):
    """BASELINE config 5 family: complex double, complex-symmetric LDLt (incl. the reference's own
    young4c.mtx fixture, 120-wide cblks)."""
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE) as p:
        p.upload(g["L0"])
        st = p.factorize(g["critere"])
        L1, _ = p.download()
    m = _lower_mask(g["cblk4"])
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", golden_names("ldlt", prec="z"))
def test_z_device_fill_matches_reference(name, golden):
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        L0, _ = p.download()
    assert np.array_equal(L0, g["L0"])


@pytest.mark.parametrize("name", golden_names("ldlh", prec="z"))
def test_z_ldlh_matches_reference_golden(name, golden):
    """z `he`: Hermitian LDL^H (reference built with -DHERMITIAN)."""
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    assert g["facto"] == 3
    with Plan(g["cblk4"], g["blok4"], 3, floattype=COMPLEXDOUBLE) as p:
        p.upload(g["L0"])
        st = p.factorize(g["critere"])
        L1, _ = p.download()
    m = _lower_mask(g["cblk4"])
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", golden_names("ldlh", prec="z") + golden_names("lu", prec="z"))
def test_z_device_fill_he_ge_matches_reference(name, golden):
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        L0, U0 = p.download()
    assert np.array_equal(L0, g["L0"])
    if g["facto"] == 2:
        assert np.array_equal(U0, g["U0"])


@pytest.mark.parametrize("name", golden_names("lu", prec="z"))
def test_z_lu_matches_reference_golden(name, golden):
    """z `ge`: complex LU with static pivoting, no conjugation."""
    from pastix_amd import COMPLEXDOUBLE
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], 2, floattype=COMPLEXDOUBLE) as p:
        p.upload(g["L0"], g["U0"])
        st = p.factorize(g["critere"])
        L1, U1 = p.download()
    scale = max(np.abs(g["L1"]).max(), np.abs(g["U1"]).max())
    assert np.abs(L1 - g["L1"]).max() <= TOL * scale
    assert np.abs(U1 - g["U1"]).max() <= TOL * scale
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", ["zrlap3d_8_ldlh", "zrlap3d_8_lu", "zrlap3d_8_ldlt"])
def test_z_one_shot_tabs(name, golden):
    """Z_{he,ge,sy}_sopalin_thread drop-ins with the reference's per-cblk interleaved complex buffers."""
    from pastix_amd.solver import sopalin_tabs
    g = golden(name)
    c4 = g["cblk4"]
    poff = np.concatenate([[0], np.cumsum(c4[:-1, 3] * (c4[:-1, 1] - c4[:-1, 0] + 1))])
    tabs = [g["L0"][poff[k]:poff[k + 1]].copy() for k in range(len(c4) - 1)]
    utabs = [g["U0"][poff[k]:poff[k + 1]].copy() for k in range(len(c4) - 1)] if g["facto"] == 2 else None
    st = sopalin_tabs(g["facto"], c4, g["blok4"], tabs, utabs, critere=g["critere"])
    L1 = np.concatenate(tabs)
    m = _lower_mask(c4) if g["facto"] != 2 else np.ones(L1.size, bool)
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * np.abs(g["L1"]).max()
    if utabs is not None:
        assert np.abs(np.concatenate(utabs) - g["U1"]).max() <= TOL * np.abs(g["U1"]).max()
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", ["rlap3d_10_llt", "rlap3d_8_ldlt", "rlap3d_8_lu", "orsirr_1030_lu",
                                  "zrlap3d_8_ldlt", "zrlap3d_8_ldlh", "zrlap3d_8_lu"])
def test_single_precision_one_shot_tabs(name, golden):
    """S_ / C_ {po,sy,he,ge}_sopalin_thread drop-ins: float panels in, float factors out (computed in f64).
    Tolerance of the single-precision path: 1e-4 * max|L| against the reference's double factors
    of the same (float-rounded) input."""
    from pastix_amd.solver import sopalin_tabs
    g = golden(name)
    c4 = g["cblk4"]
    cz = np.iscomplexobj(g["L0"])
    st_ = np.complex64 if cz else np.float32
    poff = np.concatenate([[0], np.cumsum(c4[:-1, 3] * (c4[:-1, 1] - c4[:-1, 0] + 1))])
    tabs = [g["L0"][poff[k]:poff[k + 1]].astype(st_) for k in range(len(c4) - 1)]
    utabs = [g["U0"][poff[k]:poff[k + 1]].astype(st_) for k in range(len(c4) - 1)] if g["facto"] == 2 else None
    st = sopalin_tabs(g["facto"], c4, g["blok4"], tabs, utabs, critere=g["critere"])
    assert tabs[0].dtype == st_
    L1 = np.concatenate(tabs)
    m = _lower_mask(c4) if g["facto"] in (1, 3) else recut_mask(c4) if g["facto"] == 0 else np.ones(L1.size, bool)
    assert np.abs(L1 - g["L1"])[m].max() <= 1e-4 * np.abs(g["L1"][m]).max()
    if utabs is not None:
        assert np.abs(np.concatenate(utabs) - g["U1"]).max() <= 1e-4 * np.abs(g["U1"]).max()
    assert st["nbpivot"] == g["nbpivot"]


FAST = ["rlap3d_20_llt_bs128", "rlap3d_20_lu_bs128", "zrlap3d_20_ldlt_bs128"]


@pytest.mark.parametrize("lookahead", [1024, 2048])
@pytest.mark.parametrize("name", FAST)
def test_fast_path_matches_reference_golden(name, lookahead, golden):
    """Layouts blend made with IPARM_MAX_BLOCKSIZE 128 (152-column cblks): whole 128x128 target tiles, i.e. the update
    kernel's branch-free full-tile DMA loop, tasks with several full pieces and the chunk sizes of the large
    configurations (1024 / 2048) -- compared with the reference's own factors, not with the oracle."""
    from pastix_amd import COMPLEXDOUBLE, REALDOUBLE
    g = golden(name)
    cz = np.iscomplexobj(g["L0"])
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE if cz else REALDOUBLE,
              lookahead=lookahead) as p:
        ps = p.stats()
        assert ps["full_flops"] > 0.15 * ps["update_flops"]         # whole-tile pieces exist (25 % of the update at 20^3)
        p.upload(g["L0"], g["U0"] if g["facto"] == 2 else None)
        st = p.factorize(g["critere"])
        L1, U1 = p.download()
    m = (_lower_mask(g["cblk4"]) if g["facto"] == 1 else recut_mask(g["cblk4"]) if g["facto"] == 0
         else np.ones(L1.size, bool))
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    if g["facto"] == 2:
        assert np.abs(U1 - g["U1"]).max() <= TOL * max(scale, np.abs(g["U1"]).max())
    assert st["nbpivot"] == g["nbpivot"]


QUAD = (golden_names("llt")[:4] + ["rlap3d_20_llt_bs128"] + golden_names("ldlt")[:2] + golden_names("lu")[:2] +
        ["orsirr_1030_lu"] + golden_names("ldlt", prec="z")[:2] + golden_names("ldlh", prec="z")[:1] +
        golden_names("lu", prec="z")[:1])


@pytest.mark.parametrize("fill", [25, 200])
@pytest.mark.parametrize("name", sorted(set(QUAD)))
def test_quadrant_tasks_match_reference_golden(name, fill, golden):
    """Quadrant tasks (plan.cpp) + k_update_small (kernels_small.hip): with options.quadrant_min = 1 every slot of these
    small layouts cuts its qualifying tasks into 64x64 quadrants (fill 200 %: every task without whole-tile pieces, urgent
    ones included), for every factorization kind and both arithmetics."""
    from pastix_amd import COMPLEXDOUBLE, REALDOUBLE
    g = golden(name)
    cplx = np.iscomplexobj(g["L1"])
    # (run_schedule=-1: the run launch takes whole tiles only, and on these small layouts it would take every level;
    # gather_min=-1: on these fragmented layouts the tasks would otherwise be gathered ones, which are not cut)
    with Plan(g["cblk4"], g["blok4"], g["facto"], floattype=COMPLEXDOUBLE if cplx else REALDOUBLE, quadrant_min=1,
              quadrant_fill_pct=fill, run_schedule=-1, gather_min=-1) as p:
        assert p.stats()["nquadrant_tasks"] > 0
        p.upload(g["L0"], g["U0"] if g["facto"] == 2 else None)
        st = p.factorize(g["critere"])
        L1, U1 = p.download()
    m = (_lower_mask(g["cblk4"]) if g["facto"] in (1, 3) else recut_mask(g["cblk4"]) if g["facto"] == 0
         else np.ones(g["L1"].shape, dtype=bool))
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    if g["facto"] == 2:
        assert np.abs(U1 - g["U1"]).max() <= TOL * max(scale, np.abs(g["U1"]).max())
    assert st["nbpivot"] == g["nbpivot"]


@pytest.mark.parametrize("name", ["fake_rlap3d_8_llt", "fake_rlap3d_8_ldlt", "fake_rlap3d_8_lu"])
def test_fake_fill_matches_reference(name, golden):
    """IPARM_FILL_MATRIX = API_YES (CoefMatrix_Init's structure-only fill, coefinit.c:343-443): the device fill is
    bit-exact against the input panels, the factors of the reference's own fake run are reproduced, refill re-applies
    the fill."""
    g = golden(name)
    with Plan(g["cblk4"], g["blok4"], g["facto"]) as p:
        p.fill_fake(g["n"])
        L0, U0 = p.download()
        assert np.array_equal(L0, g["L0"])
        if g["facto"] == 2:
            assert np.array_equal(U0, g["U0"])
        st = p.factorize(g["critere"])
        L1, U1 = p.download()
        p.refill()
        L0b, _ = p.download()
    assert np.array_equal(L0b, g["L0"])
    m = (_lower_mask(g["cblk4"]) if g["facto"] == 1 else recut_mask(g["cblk4"]) if g["facto"] == 0
         else np.ones(g["L1"].shape, dtype=bool))
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1 - g["L1"])[m].max() <= TOL * scale
    if g["facto"] == 2:
        assert np.abs(U1 - g["U1"]).max() <= TOL * np.abs(g["U1"]).max()
    assert st["nbpivot"] == g["nbpivot"] == 0


SINGLE = (golden_names("llt") + golden_names("ldlt") + golden_names("lu"))


@pytest.mark.parametrize("name", [n for n in SINGLE if not n.startswith("fake_")])
def test_native_single_precision_matches_reference_golden(name, golden):
    """The fp32 engine (kernels_f32.hip: k_update_s on v_mfma_f32_32x32x2_f32, k_diag_s, k_trsm_s) through the staged API
    with floattype REALSINGLE: float panels in (the reference's input rounded to float), float factors out, against the
    reference's double factors of the same input at 1e-4 * max|L| (SURVEY 8d's single-precision tolerance); static-pivot
    counts equal; the device fill from the CSC is the float rounding of the reference's fill, bit for bit."""
    from pastix_amd import REALSINGLE
    g = golden(name)
    c4 = g["cblk4"]
    lu = g["facto"] == 2
    with Plan(c4, g["blok4"], g["facto"], floattype=REALSINGLE) as p:
        p.fill_csc(g["sym"], g["n"], g["colptr"], g["rows"], g["vals"], g["perm"])
        L0, U0 = p.download()
        assert L0.dtype == np.float32 and np.array_equal(L0, g["L0"].astype(np.float32))
        if lu:
            assert np.array_equal(U0, g["U0"].astype(np.float32))
        p.upload(g["L0"], g["U0"] if lu else None)
        st = p.factorize(g["critere"])
        L1, U1 = p.download()
    m = _lower_mask(c4) if g["facto"] == 1 else recut_mask(c4) if g["facto"] == 0 else np.ones(L1.size, bool)
    scale = np.abs(g["L1"][m]).max()
    assert np.abs(L1.astype(np.float64) - g["L1"])[m].max() <= 1e-4 * scale
    if lu:
        assert np.abs(U1.astype(np.float64) - g["U1"]).max() <= 1e-4 * max(scale, np.abs(g["U1"]).max())
    assert st["nbpivot"] == g["nbpivot"]
