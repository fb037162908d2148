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
]


def main():
    only = sys.argv[1:]
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    for name, prec, kind, arg, facto, extra in CASES:
        if only and name not in only:
            continue
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_" + prec)
        raw = "/tmp/%s.bin" % name
        out = subprocess.run([exe, "dump", kind, arg, facto, "1", raw] + extra, env=env,
                             capture_output=True, text=True, check=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = fixture_io.read_raw(raw)
        fixture_io.save_npz(d, os.path.join(HERE, name + ".npz"))
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), line)
        os.remove(raw)


if __name__ == "__main__":
    main()
