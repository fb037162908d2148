"""TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

FACTO = {"llt": 0, "ldlt": 1, "lu": 2, "ldlh": 3}


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        src = [os.path.join(ROOT, "oracle", f) for f in ("sopalin_oracle.c", "sopalin_oracle_impl.h")]
        if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_fact_flops.restype = ctypes.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def fact_flops(facto, is_complex, cblk4, blok4):
    c4, b4 = _i64(cblk4), _i64(blok4)
    return lib().oracle_fact_flops(ctypes.c_int(facto), ctypes.c_int(int(is_complex)),
                                   ctypes.c_int64(len(c4) - 1), _p(c4), _p(b4))


def fill(facto, sym, n, colptr, rows, vals, perm, cblk4, blok4):
    c4, b4 = _i64(cblk4), _i64(blok4)
    cz = np.iscomplexobj(vals)
    dt = np.complex128 if cz else np.float64
    coefnbr = int(((c4[:-1, 1] - c4[:-1, 0] + 1) * c4[:-1, 3]).sum())
    L = np.zeros(coefnbr, dtype=dt)
    U = np.zeros(coefnbr, dtype=dt) if facto == 2 else None
    fn = lib().oracle_zfill if cz else lib().oracle_dfill
    v = np.ascontiguousarray(vals, dtype=dt)
    rc = fn(ctypes.c_int(facto), ctypes.c_int(int(sym)), ctypes.c_int64(n), _p(_i64(colptr)),
            _p(_i64(rows)), _p(v), _p(_i64(perm)), ctypes.c_int64(len(c4) - 1), _p(c4), _p(b4),
            _p(L), _p(U))
    assert rc == 0
    return L, U


def fill_fake(facto, gnodenbr, cblk4, cz=False):
    """coefinit.c:343-443 (IPARM_FILL_MATRIX = API_YES)."""
    c4 = _i64(cblk4)
    dt = np.complex128 if cz else np.float64
    coefnbr = int(((c4[:-1, 1] - c4[:-1, 0] + 1) * c4[:-1, 3]).sum())
    L = np.zeros(coefnbr, dtype=dt)
    U = np.zeros(coefnbr, dtype=dt) if facto == 2 else None
    fn = lib().oracle_zfill_fake if cz else lib().oracle_dfill_fake
    rc = fn(ctypes.c_int(facto), ctypes.c_int64(gnodenbr), ctypes.c_int64(len(c4) - 1), _p(c4), _p(L), _p(U))
    assert rc == 0
    return L, U


def sopalin(facto, cblk4, blok4, L, U, critere):
    """Factorize a copy of the panels; returns (L, U, nbpivot)."""
    c4, b4 = _i64(cblk4), _i64(blok4)
    cz = np.iscomplexobj(L)
    L = np.array(L, copy=True)
    U = np.array(U, copy=True) if U is not None else None
    nb = ctypes.c_int64(0)
    fn = lib().oracle_zsopalin if cz else lib().oracle_dsopalin
    rc = fn(ctypes.c_int(facto), ctypes.c_int64(len(c4) - 1), _p(c4), _p(b4), _p(L), _p(U),
            ctypes.c_double(critere), ctypes.byref(nb))
    assert rc == 0, rc
    return L, U, nb.value


def solve(facto, cblk4, blok4, L, U, b_perm):
    c4, b4 = _i64(cblk4), _i64(blok4)
    cz = np.iscomplexobj(L)
    x = np.array(b_perm, dtype=np.complex128 if cz else np.float64, copy=True)
    fn = lib().oracle_zsolve if cz else lib().oracle_dsolve
    rc = fn(ctypes.c_int(facto), ctypes.c_int64(len(c4) - 1), _p(c4), _p(b4), _p(L), _p(U), _p(x))
    assert rc == 0
    return x
