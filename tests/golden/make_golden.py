#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (development container only).

Runs oracle/_ref/ref_harness_{d,z} (built by oracle/build_ref.sh from /root/reference with
gcc + the image's MKL) and packs its raw dumps.  The fixtures are DATA (inputs + the
reference's outputs); no reference source is stored.  Fixed at 1 thread so the split is
deterministic (SURVEY 8c hazard iv).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import fixture_io  # noqa: E402

# (name, prec, kind, arg, facto, extra args)
CASES = [
    ("lap3d_6_llt", "d", "lap3d", "6", "llt", []),
    ("lap3d_8_llt", "d", "lap3d", "8", "llt", []),
    ("rlap3d_10_llt", "d", "rlap3d", "10", "llt", []),
    ("rlap3d_12_llt", "d", "rlap3d", "12", "llt", []),       # 144-wide root -> blocked (nb=64) diag path
    ("rlap3d_14_llt_bs24", "d", "rlap3d", "14", "llt", ["12", "24"]),  # small blocksizes: many splits
    ("lap1d_1000_llt", "d", "lap1d", "1000", "llt", []),      # BASELINE config 1 generator
    ("rlap3d_8_ldlt", "d", "rlap3d", "8", "ldlt", []),
    ("rlap3d_12_ldlt", "d", "rlap3d", "12", "ldlt", []),
    ("rlap3d_8_lu", "d", "rlap3d", "8", "lu", []),
    ("rlap3d_12_lu", "d", "rlap3d", "12", "lu", []),
    # complex double (BASELINE config 5 family): complex SYMMETRIC values, LDLt without conjugation
    ("zrlap3d_8_ldlt", "z", "rlap3d", "8", "ldlt", []),
    ("zrlap3d_8_lu", "z", "rlap3d", "8", "lu", []),
    ("zrlap3d_12_ldlt", "z", "rlap3d", "12", "ldlt", []),   # 144-wide root cblk: the wide-cblk complex kernels
    ("zrlap3d_12_lu", "z", "rlap3d", "12", "lu", []),
    # complex HERMITIAN values (real diagonal), LDLh: zher / TRSM "C" / GEMM "N","C"
    ("zrlap3d_8_ldlh", "z", "rlap3d", "8", "ldlh", []),
    ("zrlap3d_12_ldlh", "z", "rlap3d", "12", "ldlh", []),
    ("zyoung4c_841_ldlt", "z", "mtx", "/root/reference/src/matrix/young4c.mtx", "ldlt", []),   # the reference's own fixture
    # the reference's Harwell-Boeing fixture (oil reservoir, real unsymmetric values on a symmetric pattern): an
    # irregular, non-grid structure under the identity ordering; converted to Matrix Market for the harness
    ("orsirr_1030_lu", "d", "hb", "/root/reference/src/matrix/orsirr.rua", "lu", []),
    # IPARM_MIN/MAX_BLOCKSIZE 64/128 at 20^3: blend cuts the 609-column root into 4 cblks of 152/153 columns
    # (splitOnProcs, splitpart.c:431-475: nseq = width / max, pieces of width / nseq), so target panels have whole
    # 128x128 tiles -- the update kernel's branch-free full-tile loop and multi-piece tasks against the reference itself
    ("rlap3d_20_llt_bs128", "d", "rlap3d", "20", "llt", ["64", "128"]),
    ("rlap3d_20_lu_bs128", "d", "rlap3d", "20", "lu", ["64", "128"]),
    ("zrlap3d_20_ldlt_bs128", "z", "rlap3d", "20", "ldlt", ["64", "128"]),
    # IPARM_FILL_MATRIX = API_YES, the reference's structure-only "fake factorisation" (coefinit.c:343-443, critere
    # sopalin3d.c:597-598): the CSC is only a pattern; harness run with REF_FAKE=1.  (Names outside the
    # kind_size_facto scheme on purpose: the CSC-based tests do not apply to them.)
    ("fake_rlap3d_8_llt", "d", "rlap3d", "8", "llt", []),
    ("fake_rlap3d_8_ldlt", "d", "rlap3d", "8", "ldlt", []),
    ("fake_rlap3d_8_lu", "d", "rlap3d", "8", "lu", []),
]


def hb_to_mtx(fn, out):
    """Minimal Harwell-Boeing (assembled, real) reader -> MatrixMarket coordinate file (a data conversion)."""
    import re
    L = open(fn).read().split("\n")
    ptrc, indc, valc = [int(x) for x in L[1].split()[1:4]]
    typ = L[2][:3].upper()
    nrow, ncol, nnz = [int(x) for x in L[2][3:].split()[:3]]
    fm = re.findall(r"\(([^)]*)\)", L[3])
    width = lambda f: int(re.search(r"[IiEeDdFf](\d+)", f).group(1))
    pos = [4]

    def take(nlines, f, count, conv):
        w, vals = width(f), []
        for ln in L[pos[0]:pos[0] + nlines]:
            for i in range(0, len(ln.rstrip()), w):
                t = ln[i:i + w].strip()
                if t:
                    vals.append(conv(t.replace("D", "E").replace("d", "e")))
        pos[0] += nlines
        return vals[:count]

    ptr = take(ptrc, fm[0], ncol + 1, int)
    ind = take(indc, fm[1], nnz, int)
    val = take(valc, fm[2], nnz, float)
    with open(out, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real %s\n" % ("symmetric" if typ[1] == "S" else "general"))
        f.write("%d %d %d\n" % (nrow, ncol, nnz))
        for j in range(ncol):
            for q in range(ptr[j] - 1, ptr[j + 1] - 1):
                f.write("%d %d %.17g\n" % (ind[q], j + 1, val[q]))


def main():
    only = sys.argv[1:]
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    for name, prec, kind, arg, facto, extra in CASES:
        if only and name not in only:
            continue
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_" + prec)
        raw = "/tmp/%s.bin" % name
        if kind == "hb":
            hb_to_mtx(arg, "/tmp/%s.mtx" % name)
            kind, arg = "mtx", "/tmp/%s.mtx" % name
        out = subprocess.run([exe, "dump", kind, arg, facto, "1", raw] + extra,
                             env=dict(env, REF_FAKE="1") if name.startswith("fake_") else env,
                             capture_output=True, text=True, check=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = fixture_io.read_raw(raw)
        fixture_io.save_npz(d, os.path.join(HERE, name + ".npz"))
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), line)
        os.remove(raw)


if __name__ == "__main__":
    main()
