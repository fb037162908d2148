#!/usr/bin/env python3
"""Soak test of the fused thin-level solve (kernels.hip k_solve_thin_*: cblks synchronised by flags inside one launch):
many solves on the same factors, every one checked -- a workgroup that waited beyond the poll limit makes
pastix_amd_solve return an error, a lost contribution shows in the residual.  usage: soak_solve.py GRID FACTO REPS"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
from pastix_amd import Plan  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
facto = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
full = facto == 2
n, cp, r, v = sy.laplacian_3d(N, full=full)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm)
A = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
if not full:
    A = A + sp.tril(A, -1).T
p = Plan(s["cblk4"], s["blok4"], facto)
p.fill_csc(0 if full else 1, n, cp, r, v, s["perm"])
p.factorize(1e-14)
rng = np.random.default_rng(5)
worst, tmax, tsum = 0.0, 0.0, 0.0
pm = np.asarray(s["perm"])
for i in range(reps):
    b = rng.standard_normal(n)
    bp = np.empty(n)
    bp[pm] = b
    x = p.solve(bp)[pm]                      # (raises on any error code)
    res = float(np.linalg.norm(A @ x - b) / np.linalg.norm(b))
    dev = p.stats()["solve_time"]
    worst, tmax, tsum = max(worst, res), max(tmax, dev), tsum + dev
    assert res < 1e-10, (i, res)
print("N=%d facto=%d: %d solves, worst residual %.2e, device time mean %.2f ms, max %.2f ms" % (
    N, facto, reps, worst, tsum / reps * 1e3, tmax * 1e3))
