"""TEST INFRASTRUCTURE: read the raw dumps of oracle/ref_harness.c and the packed .npz fixtures.

A fixture holds, for one (matrix, factorization) case run through the REAL reference:
CSC + permutation, the SolverMatrix layout (cblk/blok tables), the panels right after the
reference's own fill, the panels after the reference's own factorization, and scalars.
"""
import numpy as np

FIELDS_HDR = ["n", "nnz", "facto", "prec", "cblknbr", "bloknbr", "coefnbr", "coefmax",
              "sym", "nthr", "minbs", "maxbs"]


def read_raw(path):
    """Parse the binary written by ref_harness (see its `dump` branch for the field order)."""
    buf = open(path, "rb").read()
    assert buf[:8] == b"PSTXFIX1", "bad magic"
    pos = 8

    def i64(cnt):
        nonlocal pos
        a = np.frombuffer(buf, dtype="<i8", count=cnt, offset=pos)
        pos += 8 * cnt
        return a.copy()

    hdr = dict(zip(FIELDS_HDR, (int(v) for v in i64(len(FIELDS_HDR)))))
    fdt = np.dtype("<c16") if hdr["prec"] else np.dtype("<f8")

    def flt(cnt, dt=fdt):
        nonlocal pos
        a = np.frombuffer(buf, dtype=dt, count=cnt, offset=pos)
        pos += dt.itemsize * cnt
        return a.copy()

    n, nnz = hdr["n"], hdr["nnz"]
    d = dict(hdr)
    d["colptr"] = i64(n + 1)
    d["rows"] = i64(nnz)
    d["vals"] = flt(nnz)
    d["perm"] = i64(n)
    d["invp"] = i64(n)
    d["cblk4"] = i64(4 * (hdr["cblknbr"] + 1)).reshape(-1, 4)
    d["blok4"] = i64(4 * hdr["bloknbr"]).reshape(-1, 4)
    tasknbr = int(i64(1)[0])
    d["tasks"] = i64(3 * tasknbr).reshape(-1, 3)
    nnzi = int(i64(1)[0])
    d["icsc_cnt"] = i64(n)
    d["icsc_rows"] = i64(nnzi)
    d["icsc_vals"] = flt(nnzi)
    lu = hdr["facto"] == 2
    w = d["cblk4"][:-1, 1] - d["cblk4"][:-1, 0] + 1
    sz = d["cblk4"][:-1, 3] * w
    assert int(sz.sum()) == hdr["coefnbr"]

    def panels():
        L = np.empty(hdr["coefnbr"], dtype=fdt)
        U = np.empty(hdr["coefnbr"], dtype=fdt) if lu else None
        off = 0
        for s in sz:
            s = int(s)
            L[off:off + s] = flt(s)
            if lu:
                U[off:off + s] = flt(s)
            off += s
        return L, U

    d["L0"], d["U0"] = panels()
    d["L1"], d["U1"] = panels()
    sc = flt(4, np.dtype("<f8"))
    d["critere"], d["flops"], d["time"], d["resid"] = (float(v) for v in sc)
    d["nbpivot"], d["nnzl"], d["inertia"] = (int(v) for v in i64(3))
    d["b"] = flt(n)
    d["x"] = flt(n)
    assert pos == len(buf), (pos, len(buf))
    return d


def save_npz(d, path):
    out = {k: v for k, v in d.items() if v is not None}
    np.savez_compressed(path, **out)


def load_npz(path):
    z = np.load(path)
    d = {}
    for k in z.files:
        v = z[k]
        d[k] = v.item() if v.ndim == 0 else v
    d.setdefault("U0", None)
    d.setdefault("U1", None)
    return d
