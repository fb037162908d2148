"""pastix()/iparm/dparm entry point, host-side mirror (binds pastix_amd_pastix of
include/pastix_amd_driver.h; same call sequence as src/example/src/simple.c:59-256)."""
import ctypes

import numpy as np

from . import _lib

IPARM_SIZE, DPARM_SIZE = 128, 64
# api.h:124-197 / 219-234 / 253-260
IPARM = dict(MODIFY_PARAMETER=0, START_TASK=1, END_TASK=2, VERBOSE=3, DOF_NBR=4, ITERMAX=5,
             MATRIX_VERIFICATION=6, NBITER=10, AMALGAMATION_LEVEL=13, ORDERING=14, STATIC_PIVOTING=20,
             NNZEROS=22, BASEVAL=24, MIN_BLOCKSIZE=25, MAX_BLOCKSIZE=26, SCHUR=27, FACTORIZATION=30, THREAD_NBR=34,
             LEVEL_OF_FILL=36, RHS_MAKING=38, REFINEMENT=39, SYM=40, GMRES_IM=44, FILL_MATRIX=53, INERTIA=54, FLOAT=61, ERROR_NUMBER=63,
             CUDA_NBR=64)
DPARM = dict(EPSILON_REFINEMENT=5, RELATIVE_ERROR=6, EPSILON_MAGN_CTRL=10, FACT_TIME=20, FACT_FLOPS=22)
API_TASK = dict(INIT=0, ORDERING=1, SYMBFACT=2, ANALYSE=3, NUMFACT=4, SOLVE=5, REFINE=6, CLEAN=7)
API_NO, API_YES = 0, 1
API_SYM_YES, API_SYM_NO = 0, 1
API_ORDER_SCOTCH, API_ORDER_PERSONAL = 0, 2
API_FACT_LLT, API_FACT_LDLT, API_FACT_LU = 0, 1, 2
API_RAF_GMRES, API_RAF_GRAD, API_RAF_PIVOT, API_RAF_BICGSTAB = 0, 1, 2, 3
API_REALSINGLE, API_REALDOUBLE, API_COMPLEXSINGLE, API_COMPLEXDOUBLE = 0, 1, 2, 3          # api.h:522-525
API_SYM_HER = 2
API_FACT_LDLH = 3


class PastixData:
    """Opaque pastix_data_t* handle."""

    def __init__(self):
        self.h = ctypes.c_void_p()

    def set_schur_unknown_list(self, unknowns):
        """pastix_setSchurUnknownList: unknowns (CSC base) to isolate at the end; call before the ordering task."""
        u = _lib.as_i64(unknowns)
        _lib.check(_lib.lib().pastix_amd_set_schur_unknown_list(ctypes.byref(self.h), ctypes.c_int64(len(u)), _lib.ptr(u)),
                   "pastix_amd_set_schur_unknown_list")

    def get_schur(self, nschur, dtype=np.float64):
        """pastix_getSchur: the nschur x nschur Schur complement (column-major, order of the final permutation)."""
        out = np.zeros(nschur * nschur, dtype=dtype)
        _lib.check(_lib.lib().pastix_amd_get_schur(self.h, _lib.ptr(out)), "pastix_amd_get_schur")
        return out.reshape(nschur, nschur, order="F")

    def set_grid(self, nx, ny, nz):
        _lib.check(_lib.lib().pastix_amd_set_grid(ctypes.byref(self.h), ctypes.c_int64(nx), ctypes.c_int64(ny),
                                                  ctypes.c_int64(nz)), "pastix_amd_set_grid")


def init_param():
    iparm = np.zeros(IPARM_SIZE, dtype=np.int64)
    dparm = np.zeros(DPARM_SIZE, dtype=np.float64)
    iparm[IPARM["MODIFY_PARAMETER"]] = API_NO
    pastix(None, 0, None, None, None, None, None, None, 1, iparm, dparm)
    return iparm, dparm


def pastix(pastix_data, n, colptr, rows, avals, perm, invp, b, nrhs, iparm, dparm):
    """void pastix(pastix_data_t**, MPI_Comm, n, colptr, row, avals, perm, invp, b, rhs, iparm, dparm)
    (pastix.h:219-222).  Arrays are int64 / float64 numpy arrays (complex128 / float32 / complex64 values and right-hand
    sides with iparm[IPARM_FLOAT] = API_COMPLEXDOUBLE / API_REALSINGLE / API_COMPLEXSINGLE), modified in place."""
    pd = pastix_data if pastix_data is not None else PastixData()
    L = _lib.lib()
    L.pastix_amd_pastix.restype = None
    L.pastix_amd_pastix(ctypes.byref(pd.h), 0, ctypes.c_int64(n), _lib.ptr(colptr), _lib.ptr(rows),
                        _lib.ptr(avals), _lib.ptr(perm), _lib.ptr(invp), _lib.ptr(b), ctypes.c_int64(nrhs),
                        _lib.ptr(iparm), _lib.ptr(dparm))
    return pd
