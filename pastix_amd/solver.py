"""Host-side handle on the device engine: plan once, fill / upload, factorize, download.

Mirrors the life cycle around the reference's sopalin step (pastix_task_sopalin,
src/sopalin/src/pastix.c:3439-3956): coefficient fill -> {po,sy,ge}_sopalin_thread -> factors
left in the panels for the solve.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import LayoutArrays, Options, Stats, check

FACT_LLT, FACT_LDLT, FACT_LU, FACT_LDLH = 0, 1, 2, 3
REALSINGLE, REALDOUBLE, COMPLEXDOUBLE = 0, 1, 3        # IPARM_FLOAT values (api.h:522-525)


def fact_flops(cblk4, blok4, factotype=FACT_LLT, floattype=REALDOUBLE):
    la = LayoutArrays(cblk4, blok4)
    return _lib.lib().pastix_amd_fact_flops(ctypes.byref(la.c), factotype, floattype)


class Plan:
    def __init__(self, cblk4, blok4, factotype=FACT_LLT, floattype=REALDOUBLE, device=0, lookahead=0, verbose=0,
                 schur=False, quadrant_min=0, quadrant_fill_pct=0, run_schedule=0, run_max_cblks=0, run_t_workers=0, run_d_workers=0, gather_min=0):
        self.layout = LayoutArrays(cblk4, blok4)
        self.factotype = factotype
        # panels / CSC values of the plan's arithmetic; the vectors of a solve stay double for single-precision plans
        self.dtype = np.complex128 if floattype == COMPLEXDOUBLE else np.float32 if floattype == REALSINGLE else np.float64
        self.vdtype = np.complex128 if floattype == COMPLEXDOUBLE else np.float64
        self._h = ctypes.c_void_p()
        opts = Options()
        opts.device = device
        opts.lookahead = lookahead
        opts.verbose = verbose
        opts.schur = 1 if schur else 0
        opts.quadrant_min = quadrant_min
        opts.quadrant_fill_pct = quadrant_fill_pct
        opts.run_schedule = run_schedule
        opts.run_max_cblks = run_max_cblks
        opts.run_t_workers = run_t_workers
        opts.run_d_workers = run_d_workers
        opts.gather_min = gather_min
        check(_lib.lib().pastix_amd_plan_create(ctypes.byref(self.layout.c), factotype, floattype,
                                                ctypes.byref(opts), ctypes.byref(self._h)),
              "pastix_amd_plan_create")

    def close(self):
        if self._h:
            _lib.lib().pastix_amd_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def coefnbr(self):
        return self.layout.coefnbr()

    def stats(self):
        s = Stats()
        check(_lib.lib().pastix_amd_plan_stats(self._h, ctypes.byref(s)), "pastix_amd_plan_stats")
        return s.as_dict()

    def upload(self, L, U=None):
        L = np.ascontiguousarray(L, dtype=self.dtype)
        assert L.size == self.coefnbr
        U = np.ascontiguousarray(U, dtype=self.dtype) if U is not None else None
        check(_lib.lib().pastix_amd_upload_packed(self._h, _lib.ptr(L), _lib.ptr(U)), "pastix_amd_upload_packed")

    def download(self):
        L = np.empty(self.coefnbr, dtype=self.dtype)
        U = np.empty(self.coefnbr, dtype=self.dtype) if self.factotype == FACT_LU else None
        check(_lib.lib().pastix_amd_download_packed(self._h, _lib.ptr(L), _lib.ptr(U)), "pastix_amd_download_packed")
        return L, U

    def fill_csc(self, sym, n, colptr, rows, vals, perm):
        colptr, rows, perm = _lib.as_i64(colptr), _lib.as_i64(rows), _lib.as_i64(perm)
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        check(_lib.lib().pastix_amd_fill_csc(self._h, int(sym), ctypes.c_int64(n), _lib.ptr(colptr),
                                             _lib.ptr(rows), _lib.ptr(vals), _lib.ptr(perm)),
              "pastix_amd_fill_csc")

    def fill_fake(self, gnodenbr):
        """IPARM_FILL_MATRIX = API_YES: the reference's structure-only fill (coefinit.c:343-443)."""
        check(_lib.lib().pastix_amd_fill_fake(self._h, ctypes.c_int64(gnodenbr)), "pastix_amd_fill_fake")

    def refill(self):
        check(_lib.lib().pastix_amd_refill(self._h), "pastix_amd_refill")

    def factorize(self, critere, allow_numeric_error=False):
        s = Stats()
        rc = _lib.lib().pastix_amd_factorize(self._h, ctypes.c_double(critere), ctypes.byref(s))
        if rc != 0 and not (allow_numeric_error and rc == -4):
            check(rc, "pastix_amd_factorize")
        d = s.as_dict()
        d["rc"] = rc
        return d

    def solve(self, x):
        """x (permuted numbering; n or n x nrhs) -> solution.  A contiguous 1-D array of the plan's dtype is solved in
        place, like the reference's b; anything else is copied."""
        x = np.ascontiguousarray(x, dtype=self.vdtype)     # complex plans: interleaved complex128, like the reference
        nrhs = 1 if x.ndim == 1 else x.shape[1]
        xf = np.asfortranarray(x.reshape(len(x), nrhs))
        check(_lib.lib().pastix_amd_solve(self._h, _lib.ptr(xf), ctypes.c_int64(nrhs)), "pastix_amd_solve")
        return xf.reshape(x.shape) if x.ndim == 1 else np.ascontiguousarray(xf)


def sopalin_tabs(factotype, cblk4, blok4, coeftab, ucoeftab=None, critere=0.0, lookahead=0):
    """One-shot drop-in call {po,sy,ge,he}_sopalin with the reference's per-cblk host buffers
    (lists of 1-D float64 / complex128 / float32 / complex64 arrays -- the D_ / Z_ / S_ / C_ variants --, factorized in
    place; the single-precision variants compute in f64)."""
    la = LayoutArrays(cblk4, blok4)
    n = la.cblknbr
    arr = (ctypes.c_void_p * n)(*[a.ctypes.data for a in coeftab])
    opts = Options()
    opts.lookahead = lookahead
    s = Stats()
    L = _lib.lib()
    dt = coeftab[0].dtype if coeftab else np.dtype(np.float64)
    if dt in (np.complex128, np.complex64):
        pre = "z" if dt == np.complex128 else "c"          # Z_ or C_ {sy,he,ge}_sopalin_thread
        cr = ctypes.c_double(critere)
        if factotype == FACT_LU:
            uarr = (ctypes.c_void_p * n)(*[a.ctypes.data for a in ucoeftab])
            rc = getattr(L, "pastix_amd_%s_ge_sopalin" % pre)(ctypes.byref(la.c), arr, uarr, cr, ctypes.byref(opts), ctypes.byref(s))
        elif factotype == FACT_LDLT:
            rc = getattr(L, "pastix_amd_%s_sy_sopalin" % pre)(ctypes.byref(la.c), arr, cr, ctypes.byref(opts), ctypes.byref(s))
        elif factotype == 3:
            rc = getattr(L, "pastix_amd_%s_he_sopalin" % pre)(ctypes.byref(la.c), arr, cr, ctypes.byref(opts), ctypes.byref(s))
        else:
            rc = -5
        check(rc, "pastix_amd_%s_*_sopalin" % pre)
        return s.as_dict()
    if dt == np.float32:                                   # S_ {po,sy,ge}_sopalin_thread
        cr = ctypes.c_double(critere)
        if factotype == FACT_LLT:
            rc = L.pastix_amd_s_po_sopalin(ctypes.byref(la.c), arr, cr, ctypes.byref(opts), ctypes.byref(s))
        elif factotype == FACT_LDLT:
            rc = L.pastix_amd_s_sy_sopalin(ctypes.byref(la.c), arr, cr, ctypes.byref(opts), ctypes.byref(s))
        else:
            uarr = (ctypes.c_void_p * n)(*[a.ctypes.data for a in ucoeftab])
            rc = L.pastix_amd_s_ge_sopalin(ctypes.byref(la.c), arr, uarr, cr, ctypes.byref(opts), ctypes.byref(s))
        check(rc, "pastix_amd_s_*_sopalin")
        return s.as_dict()
    if factotype == FACT_LLT:
        rc = L.pastix_amd_d_po_sopalin(ctypes.byref(la.c), arr, ctypes.c_double(critere), ctypes.byref(opts), ctypes.byref(s))
    elif factotype == FACT_LDLT:
        rc = L.pastix_amd_d_sy_sopalin(ctypes.byref(la.c), arr, ctypes.c_double(critere), ctypes.byref(opts), ctypes.byref(s))
    else:
        uarr = (ctypes.c_void_p * n)(*[a.ctypes.data for a in ucoeftab])
        rc = L.pastix_amd_d_ge_sopalin(ctypes.byref(la.c), arr, uarr, ctypes.c_double(critere), ctypes.byref(opts), ctypes.byref(s))
    check(rc, "pastix_amd_d_*_sopalin")
    return s.as_dict()
