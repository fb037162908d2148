"""ctypes binding of pastix_amd/lib/libpastix_amd.so (the C ABI of include/pastix_amd.h).

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible the
product path raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from this package.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PASTIX_AMD_LIB: another build of the same library, for A/B timing of compile-time switches)
LIB_PATH = os.environ.get("PASTIX_AMD_LIB") or os.path.join(_HERE, "lib", "libpastix_amd.so")

i64 = ctypes.c_int64


class CBlk(ctypes.Structure):
    _fields_ = [("fcolnum", i64), ("lcolnum", i64), ("bloknum", i64), ("stride", i64)]


class Blok(ctypes.Structure):
    _fields_ = [("frownum", i64), ("lrownum", i64), ("cblknum", i64), ("coefind", i64)]


class Layout(ctypes.Structure):
    _fields_ = [("cblknbr", i64), ("bloknbr", i64),
                ("cblktab", ctypes.c_void_p), ("bloktab", ctypes.c_void_p)]


class Options(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("lookahead", ctypes.c_int), ("verbose", ctypes.c_int),
                ("external_arena", ctypes.c_int), ("schur", ctypes.c_int), ("quadrant_min", ctypes.c_int),
                ("quadrant_fill_pct", ctypes.c_int), ("run_schedule", ctypes.c_int), ("run_max_cblks", ctypes.c_int),
                ("run_t_workers", ctypes.c_int), ("run_d_workers", ctypes.c_int), ("gather_min", ctypes.c_int),
                ("reserved", ctypes.c_int * 4)]


class Stats(ctypes.Structure):
    _fields_ = [("fact_flops", ctypes.c_double), ("fact_time", ctypes.c_double),
                ("update_time", ctypes.c_double), ("h2d_time", ctypes.c_double),
                ("d2h_time", ctypes.c_double), ("nbpivot", i64), ("coefnbr", i64),
                ("nlevels", i64), ("ntasks", i64), ("npieces", i64), ("nupdate_launches", i64), ("inertia", i64),
                ("update_flops", ctypes.c_double), ("local_flops", ctypes.c_double),
                ("update_bytes", ctypes.c_double), ("full_flops", ctypes.c_double),
                ("update_time_sum", ctypes.c_double), ("urgent_flops", ctypes.c_double),
                ("urgent_time_sum", ctypes.c_double), ("nurgent_launches", ctypes.c_int64),
                ("solve_time", ctypes.c_double), ("nquadrant_tasks", ctypes.c_double),
                ("run_time", ctypes.c_double), ("run_flops", ctypes.c_double), ("run_tickets", i64), ("run_first_level", i64),
                ("plan_time", ctypes.c_double), ("total_time", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


ERRORS = {0: "OK", -1: "BADPARAMETER", -2: "ALLOC", -3: "DEVICE", -4: "NUMERIC", -5: "UNSUPPORTED",
          -6: "LAYOUT", -7: "TIMEOUT"}


class PastixAmdError(RuntimeError):
    def __init__(self, code, where):
        super().__init__("%s failed: %s (%d)" % (where, ERRORS.get(code, "?"), code))
        self.code = code


_lib = None

# every symbol include/pastix_amd.h declares
EXPORTS = [
    "pastix_amd_release_cached_plan", "pastix_amd_plan_check_pieces", "pastix_amd_plan_run_edges_digest", "pastix_amd_d_po_sopalin", "pastix_amd_d_sy_sopalin", "pastix_amd_d_ge_sopalin", "pastix_amd_z_sy_sopalin",
    "pastix_amd_z_he_sopalin", "pastix_amd_z_ge_sopalin",
    "pastix_amd_s_po_sopalin", "pastix_amd_s_sy_sopalin", "pastix_amd_s_ge_sopalin",
    "pastix_amd_c_sy_sopalin", "pastix_amd_c_he_sopalin", "pastix_amd_c_ge_sopalin",
    "pastix_amd_plan_create", "pastix_amd_plan_destroy", "pastix_amd_plan_stats", "pastix_amd_dist_schedule_hash",
    "pastix_amd_upload_packed", "pastix_amd_download_packed", "pastix_amd_upload_tabs",
    "pastix_amd_download_tabs", "pastix_amd_fill_csc", "pastix_amd_refill", "pastix_amd_factorize", "pastix_amd_solve", "pastix_amd_solve_device", "pastix_amd_refine",
    "pastix_amd_device_arenas", "pastix_amd_fact_flops", "pastix_amd_version",
    "pastix_amd_plan_create_dist", "pastix_amd_plan_layout_info", "pastix_amd_plan_set_arena", "pastix_amd_plan_arena_info", "pastix_amd_fill_fake",
    "pastix_amd_plan_set_stream", "pastix_amd_factorize_begin", "pastix_amd_factorize_level",
    "pastix_amd_factorize_end", "pastix_amd_plan_profile", "pastix_amd_plan_run_info", "pastix_amd_fanin_touched", "pastix_amd_plan_fanin_add", "pastix_amd_download_cblk",
    "pastix_amd_dist_unique_id", "pastix_amd_dist_selftest_rccl", "pastix_amd_dist_attach_rccl", "pastix_amd_dist_attach_local", "pastix_amd_dist_info",
    "pastix_amd_dist_partition", "pastix_amd_factorize_dist", "pastix_amd_factorize_dist_local", "pastix_amd_solve_dist", "pastix_amd_solve_dist_local", "pastix_amd_dist_schedule",
]
# include/pastix_amd_symbolic.h and include/pastix_amd_driver.h
EXPORTS_HOST = [
    "pastix_amd_order_grid", "pastix_amd_order_graph", "pastix_amd_symbolic", "pastix_amd_symbol_layout", "pastix_amd_symbol_perm",
    "pastix_amd_symbol_info", "pastix_amd_symbol_destroy", "pastix_amd_pastix", "pastix_amd_set_grid", "pastix_amd_set_schur_unknown_list", "pastix_amd_get_schur",
    "pastix_amd_data_plan",
]


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  If this library pulls in the system copy first and
    torch is imported later, the process ends up with two HIP runtimes and torch reports "No HIP GPUs are
    available".  Loading torch's copy first (without importing torch) makes both use one runtime, whatever the
    import order; without torch installed nothing happens and the system runtime is used."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001  (best effort: the system runtime still works on its own)
        pass


def share_rccl_with_torch():
    """Same for RCCL: the multi-GPU driver resolves librccl at run time (csrc/dist.cpp) and takes a copy that is already
    in the process.  PyTorch's wheel bundles its own librccl.so next to the HIP runtime loaded above; the ROCm
    installation's copy beside that runtime corrupts the heap at exit.  Call before the first RCCL use."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("pastix_amd: %s is missing; run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(LIB_PATH)
        L.pastix_amd_fact_flops.restype = ctypes.c_double
        L.pastix_amd_version.restype = ctypes.c_char_p
        L.pastix_amd_plan_destroy.restype = None
        _lib = L
    return _lib


def check(code, where):
    if code != 0:
        raise PastixAmdError(code, where)


def as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class LayoutArrays:
    """Keeps the numpy arrays alive that a Layout struct points into."""

    def __init__(self, cblk4, blok4):
        self.cblk4 = as_i64(cblk4).reshape(-1, 4)
        self.blok4 = as_i64(blok4).reshape(-1, 4)
        self.c = Layout(len(self.cblk4) - 1, len(self.blok4), ptr(self.cblk4), ptr(self.blok4))

    @property
    def cblknbr(self):
        return len(self.cblk4) - 1

    def widths(self):
        return self.cblk4[:-1, 1] - self.cblk4[:-1, 0] + 1

    def panel_offsets(self):
        sz = self.widths() * self.cblk4[:-1, 3]
        return np.concatenate([[0], np.cumsum(sz)]).astype(np.int64)

    def coefnbr(self):
        return int(self.panel_offsets()[-1])
