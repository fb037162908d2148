"""The reference as the CALLER of the MI355X engine (north_star: "the host-side C driver and the symbolic pipeline stay
as-is and hand the SolverMatrix across a thin C-ABI").

oracle/_ref/ref_harness_{d,z}_ob_amd (built by oracle/build_ref.sh in the development container, travels with gpurun)
is the REAL PaStiX 5.2.2.16 -- pastix(), kass, blend, CoefMatrix_Init, updo -- in which the numerical factorization call
of sopalin_thread() (src/sopalin/src/sopalin3d.c:1411) is bound to integration/sopalin_amd_stub.h -> libpastix_amd.so.
Mode `amd` runs the factorization on the GPU, mode `time` of the same binary on the reference's CPU engine; both then
solve with the reference's own updo and report ||Ax-b||/||b||, IPARM_STATIC_PIVOTING, IPARM_INERTIA.
Skipped where oracle/_ref was not built (a clean clone without /root/reference)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REF = os.path.join(ROOT, "oracle", "_ref")


def _run(prec, mode, kind, arg, facto, extra=()):
    exe = os.path.join(REF, "ref_harness_%s_ob_amd" % prec)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_harness_%s_ob_amd not built (needs /root/reference at build time)" % prec)
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1")
    env.pop("PASTIX_AMD_ENGINE", None)
    out = subprocess.run([exe, mode, kind, str(arg), facto, "1", "/dev/null"] + [str(x) for x in extra], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def _write_mtx(g, path):
    """Matrix Market file of a golden fixture's CSC (data the reference's own fixture files hold)."""
    cp, rows, vals = g["colptr"], g["rows"], g["vals"]
    n = int(g["n"])
    cz = np.iscomplexobj(vals)
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n" % ("complex" if cz else "real",
                                                                 "symmetric" if g["sym"] else "general"))
        f.write("%d %d %d\n" % (n, n, len(rows)))
        for j in range(n):
            for q in range(cp[j] - 1, cp[j + 1] - 1):
                if cz:
                    f.write("%d %d %.17g %.17g\n" % (rows[q], j + 1, vals[q].real, vals[q].imag))
                else:
                    f.write("%d %d %.17g\n" % (rows[q], j + 1, vals[q]))


@pytest.mark.parametrize("prec,kind,arg,facto,extra", [
    ("d", "rlap3d", 20, "llt", ()),                # north_star's case: 20^3 through pastix() -> blend -> GPU -> updo
    ("d", "rlap3d", 20, "llt", (64, 128)),         # blend with MAX_BLOCKSIZE 128: whole 128x128 tiles
    ("d", "lap1d", 1000, "llt", ()),               # BASELINE config 1's generator
    ("d", "rlap3d", 16, "ldlt", ()),
    ("d", "rlap3d", 16, "lu", ()),
    ("z", "rlap3d", 12, "ldlt", ()),
    ("z", "rlap3d", 12, "ldlh", ()),
    ("z", "rlap3d", 10, "lu", ()),
])
def test_reference_pastix_drives_the_gpu_engine(prec, kind, arg, facto, extra):
    gpu = _run(prec, "amd", kind, arg, facto, extra)
    cpu = _run(prec, "time", kind, arg, facto, extra)
    assert gpu["gpu_engine_calls"] == 1 and gpu["gpu_engine_rc"] == 0      # the factorization ran on the MI355X engine
    assert cpu["gpu_engine_calls"] == 0
    assert gpu["residual"] <= 1e-10                                         # reference updo on the GPU's factors
    assert gpu["static_pivots"] == cpu["static_pivots"]                     # IPARM_STATIC_PIVOTING
    assert gpu["inertia"] == cpu["inertia"]                                 # IPARM_INERTIA (real LDLt: n, else -1)
    assert gpu["flops"] == cpu["flops"] and gpu["nnzl"] == cpu["nnzl"]


@pytest.mark.parametrize("prec,name,facto", [("z", "zyoung4c_841_ldlt", "ldlt"), ("d", "orsirr_1030_lu", "lu")])
def test_reference_fixture_matrices_through_the_gpu_engine(prec, name, facto, golden, tmp_path):
    """The reference's own matrix files (young4c.mtx: complex symmetric 841^2; orsirr.rua: real unsymmetric 1030^2 with
    a 334-column supernode), re-written from the committed fixtures."""
    g = golden(name)
    path = str(tmp_path / (name + ".mtx"))
    _write_mtx(g, path)
    gpu = _run(prec, "amd", "mtx", path, facto)
    cpu = _run(prec, "time", "mtx", path, facto)
    assert gpu["gpu_engine_calls"] == 1 and gpu["gpu_engine_rc"] == 0
    assert gpu["residual"] <= max(1e-10, 10 * cpu["residual"])
    assert gpu["static_pivots"] == cpu["static_pivots"] == g["nbpivot"]
    assert gpu["cblknbr"] == len(g["cblk4"]) - 1                            # the layout blend made is the fixture's


def _cmp(prec, kind, arg, facto, extra=(), threads=32, contig=True, timeout=1500):
    exe = os.path.join(REF, "ref_harness_%s_ob_amd" % prec)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_harness_%s_ob_amd not built (needs /root/reference at build time)" % prec)
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1")
    env.pop("PASTIX_AMD_ENGINE", None)
    if contig:
        env["REF_ORDER_CONTIG"] = "1"
    ncpu = os.cpu_count() or 1
    out = subprocess.run([exe, "cmp", kind, str(arg), facto, str(max(1, min(threads, ncpu))), "/dev/null"] + [str(x) for x in extra],
                         env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    js = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    c = [j for j in js if j.get("cmp")][0]
    return c, js[-1]


# PARITY AT SCALE (the fixture files stop at 20^3): ONE analysis by the real kass + blend, the numerical factorization
# twice on the same SolverMatrix -- the reference's CPU engine (its threads, its BLAS) and the MI355X engine through the
# stub -- and the factors compared entry by entry inside the harness (oracle/ref_harness.c, mode cmp; sopalin3d.c:1388-1422
# is the call that is swapped).  These sizes run the schedule of the big configurations on blend's own layouts: chunks of
# 1024, tasks of up to 8 / 16 pieces, quadrant tasks at their production thresholds, re-cut cblks (blend leaves pieces of
# 144 / 152 columns), the run launch over hundreds of thin levels.
@pytest.mark.parametrize("prec,arg,facto,extra", [
    ("d", 60, "llt", ()),
    ("d", 60, "llt", (64, 128)),
    ("d", 80, "llt", ()),
    ("d", 60, "ldlt", ()),
    ("d", 80, "ldlt", (64, 128)),
    ("d", 60, "lu", ()),
    ("d", 80, "lu", (64, 128)),
    ("z", 24, "ldlt", ()),
    ("z", 32, "ldlt", (64, 128)),
    ("z", 24, "lu", ()),
    # the sizes of the configurations themselves (configs[1]: 100^3 dLLt; configs[2]'s factorization at the size one host run
    # of the reference allows: 100^3 dLU; configs[4]: z LDLt): 250 k tickets of the run launch against the reference
    ("d", 100, "llt", (64, 128)),
    ("d", 100, "lu", (64, 128)),
    ("z", 40, "ldlt", (64, 128)),
])
def test_factors_equal_the_reference_cpu_engine_at_scale(prec, arg, facto, extra):
    c, last = _cmp(prec, "rlap3d", arg, facto, extra)
    assert c["gpu_engine_calls"] == 1 and c["gpu_engine_rc"] == 0
    assert c["rel_L"] <= 1e-12, c                      # max |L_gpu - L_ref| / max |L_ref|  (SURVEY 8d tolerance)
    if facto == "lu":
        assert c["rel_U"] <= 1e-12, c
    assert c["static_pivots_gpu"] == c["static_pivots_ref"]
    assert c["inertia_gpu"] == c["inertia_ref"]
    assert last["residual"] <= 1e-10                   # the reference's updo on the GPU's factors


# The same comparison on FRAGMENTED layouts: the harness's default numbering of the separator nodes (lexicographic) gives
# blend bloks of 2-4 rows every 50-60 -- 2.9 x the bloks of the contiguous numbering --, which the plan turns into GATHERED
# pieces (plan.cpp; round 5).  Real and complex arithmetic, every factorization that gathers.
@pytest.mark.parametrize("prec,arg,facto,extra", [
    ("d", 60, "llt", ()),
    ("d", 60, "ldlt", (64, 128)),
    ("d", 60, "lu", (64, 128)),
    ("z", 24, "ldlt", ()),
])
def test_gathered_pieces_equal_the_reference_cpu_engine(prec, arg, facto, extra):
    c, last = _cmp(prec, "rlap3d", arg, facto, extra, contig=False)
    assert c["gpu_engine_calls"] == 1 and c["gpu_engine_rc"] == 0
    assert c["rel_L"] <= 1e-12, c
    if facto == "lu":
        assert c["rel_U"] <= 1e-12, c
    assert c["static_pivots_gpu"] == c["static_pivots_ref"]
    assert c["inertia_gpu"] == c["inertia_ref"]
    assert last["residual"] <= 1e-10
