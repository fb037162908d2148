"""Host-side layout producer binding (include/pastix_amd_symbolic.h): ordering for grids and
symbolic factorization -> cblk/blok tables in the SolverMatrix data model."""
import ctypes

import numpy as np

from . import _lib
from ._lib import Layout, check


class SymOptions(ctypes.Structure):
    _fields_ = [("max_blocksize", ctypes.c_int), ("amalgamation_pct", ctypes.c_int),
                ("max_merge_width", ctypes.c_int), ("schur_n", ctypes.c_int), ("blend_split", ctypes.c_int),
                ("min_blocksize", ctypes.c_int), ("candidate_procs", ctypes.c_int), ("reserved", ctypes.c_int * 9)]


def order_grid(nx, ny, nz, leaf=8):
    n = nx * ny * nz
    perm = np.empty(n, dtype=np.int64)
    invp = np.empty(n, dtype=np.int64)
    check(_lib.lib().pastix_amd_order_grid(ctypes.c_int64(nx), ctypes.c_int64(ny), ctypes.c_int64(nz),
                                           int(leaf), _lib.ptr(perm), _lib.ptr(invp)), "pastix_amd_order_grid")
    return perm, invp


def order_graph(n, colptr, rows, leaf=64):
    """Nested dissection of a general graph (pastix_amd_order_graph): the fallback ordering of the pastix() driver."""
    colptr, rows = _lib.as_i64(colptr), _lib.as_i64(rows)
    perm = np.empty(n, dtype=np.int64)
    invp = np.empty(n, dtype=np.int64)
    check(_lib.lib().pastix_amd_order_graph(ctypes.c_int64(n), _lib.ptr(colptr), _lib.ptr(rows), int(leaf),
                                            _lib.ptr(perm), _lib.ptr(invp)), "pastix_amd_order_graph")
    return perm, invp


def symbolic(n, colptr, rows, perm=None, max_blocksize=128, amalgamation_pct=5, max_merge_width=0, schur_n=0,
             blend_split=False, min_blocksize=0, candidate_procs=1):
    """Returns dict(perm, invp, cblk4, blok4, nnzl, nsuper_fund, nsuper_amalg)."""
    colptr, rows = _lib.as_i64(colptr), _lib.as_i64(rows)
    perm = _lib.as_i64(perm) if perm is not None else None
    o = SymOptions()
    o.max_blocksize, o.amalgamation_pct, o.max_merge_width = int(max_blocksize), int(amalgamation_pct), int(max_merge_width)
    o.schur_n = int(schur_n)
    o.blend_split, o.min_blocksize, o.candidate_procs = int(bool(blend_split)), int(min_blocksize), int(candidate_procs)
    h = ctypes.c_void_p()
    L = _lib.lib()
    check(L.pastix_amd_symbolic(ctypes.c_int64(n), _lib.ptr(colptr), _lib.ptr(rows), _lib.ptr(perm),
                                ctypes.byref(o), ctypes.byref(h)), "pastix_amd_symbolic")
    try:
        lay = Layout()
        check(L.pastix_amd_symbol_layout(h, ctypes.byref(lay)), "pastix_amd_symbol_layout")
        c4 = np.ctypeslib.as_array(ctypes.cast(lay.cblktab, ctypes.POINTER(ctypes.c_int64)),
                                   shape=(lay.cblknbr + 1, 4)).copy()
        b4 = np.ctypeslib.as_array(ctypes.cast(lay.bloktab, ctypes.POINTER(ctypes.c_int64)),
                                   shape=(lay.bloknbr, 4)).copy()
        pp, ip = ctypes.POINTER(ctypes.c_int64)(), ctypes.POINTER(ctypes.c_int64)()
        check(L.pastix_amd_symbol_perm(h, ctypes.byref(pp), ctypes.byref(ip)), "pastix_amd_symbol_perm")
        p = np.ctypeslib.as_array(pp, shape=(n,)).copy()
        i = np.ctypeslib.as_array(ip, shape=(n,)).copy()
        info = np.zeros(8, dtype=np.int64)
        check(L.pastix_amd_symbol_info(h, _lib.ptr(info)), "pastix_amd_symbol_info")
    finally:
        L.pastix_amd_symbol_destroy(ctypes.c_void_p(h.value))
    return dict(perm=p, invp=i, cblk4=c4, blok4=b4, nnzl=int(info[3]), nsuper_fund=int(info[4]),
                nsuper_amalg=int(info[5]))


def laplacian_3d(nx, ny=None, nz=None, full=False):
    """3-D 7-point Laplacian (BASELINE config 2: diag 6, off-diagonal -1, Dirichlet truncation),
    node id = x + nx*(y + ny*z).  CSC 1-based; lower triangle unless full."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    n = nx * ny * nz
    ids = np.arange(n, dtype=np.int64)
    x = ids % nx
    y = (ids // nx) % ny
    z = ids // (nx * ny)
    cols = [ids]
    rws = [ids]
    vals = [np.full(n, 6.0)]
    for m, d in ((x < nx - 1, 1), (y < ny - 1, nx), (z < nz - 1, nx * ny)):
        cols.append(ids[m]); rws.append(ids[m] + d); vals.append(np.full(int(m.sum()), -1.0))
        if full:
            cols.append(ids[m] + d); rws.append(ids[m]); vals.append(np.full(int(m.sum()), -1.0))
    cols = np.concatenate(cols); rws = np.concatenate(rws); vals = np.concatenate(vals)
    order = np.lexsort((rws, cols))
    cols, rws, vals = cols[order], rws[order], vals[order]
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, cols + 1, 1)
    colptr = np.cumsum(colptr) + 1
    return n, colptr, rws + 1, vals


def elasticity_3d(N, dof=3, seed=12345):
    """BASELINE config 5 matrix: 3 dof per node on an N^3 grid with 7-point node coupling, every node
    pair coupled by a full dof x dof block; values complex SYMMETRIC (not Hermitian), off-diagonal
    entries (u + i v), u,v ~ U(-1,1) (seeded), diagonal = 1 + sum |off-diagonals of the row| (real).
    Returns (n, colptr, rows, vals, node_of_dof): lower-triangular CSC, 1-based, complex128."""
    rng = np.random.default_rng(seed)
    nn = N * N * N
    ids = np.arange(nn, dtype=np.int64)
    x, y, z = ids % N, (ids // N) % N, ids // (N * N)
    pairs_i, pairs_j = [ids], [ids]                 # node pairs (i >= j): self + 3 forward neighbours
    for m, d in ((x < N - 1, 1), (y < N - 1, N), (z < N - 1, N * N)):
        pairs_i.append(ids[m] + d)
        pairs_j.append(ids[m])
    pi, pj = np.concatenate(pairs_i), np.concatenate(pairs_j)
    rows, cols = [], []
    for a in range(dof):
        for b in range(dof):
            r_, c_ = pi * dof + a, pj * dof + b
            keep = r_ >= c_                          # lower triangle (drops the upper part of self blocks)
            rows.append(r_[keep])
            cols.append(c_[keep])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    order = np.lexsort((rows, cols))
    rows, cols = rows[order], cols[order]
    n = nn * dof
    vals = rng.uniform(-1, 1, len(rows)) + 1j * rng.uniform(-1, 1, len(rows))
    offd = rows != cols
    rowsum = np.zeros(n)
    np.add.at(rowsum, rows[offd], np.abs(vals[offd]))
    np.add.at(rowsum, cols[offd], np.abs(vals[offd]))
    vals[~offd] = rowsum[rows[~offd]] + 1.0
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, cols + 1, 1)
    colptr = np.cumsum(colptr) + 1
    return n, colptr, rows + 1, vals.astype(np.complex128), np.arange(n) // dof


def order_grid_dof(N, dof, leaf=8):
    """Geometric ND of the node grid, expanded to dofs (all dofs of a node stay adjacent)."""
    pn, _ = order_grid(N, N, N, leaf=leaf)
    perm = (pn[:, None] * dof + np.arange(dof)[None, :]).reshape(-1)
    invp = np.empty_like(perm)
    invp[perm] = np.arange(len(perm))
    return perm, invp
