#!/usr/bin/env python3
"""Developer benchmark of the device solve (forward + backward sweep over the factored panels):
wall time per solve (host vector in, host vector out), GB/s over the panel bytes, residual."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
from pastix_amd import Plan  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=100)
ap.add_argument("--facto", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nrhs", type=int, default=1)
a = ap.parse_args()
N = a.n
n, cp, r, v = sy.laplacian_3d(N)
perm, invp = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm)
c4, b4 = s["cblk4"], s["blok4"]
p = Plan(c4, b4, a.facto)
p.fill_csc(1, n, cp, r, v, s["perm"])
st = p.factorize(1e-14)
Al = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
A = Al + sp.tril(Al, -1).T
rng = np.random.default_rng(1)
xs = rng.standard_normal((n, a.nrhs)) if a.nrhs > 1 else rng.standard_normal(n)
b = A @ xs
pm = np.asarray(s["perm"])
nbytes = 8.0 * s["nnzl"] * (2 if a.facto != 2 else 2)        # both sweeps read the panels once
for rep in range(a.reps):
    bp = np.empty(b.shape)
    bp[pm] = b
    t = time.time()
    xp = p.solve(bp)
    dt = time.time() - t
    x = xp[pm]
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    dev = p.stats()["solve_time"]
    print("N=%d facto=%d nrhs=%d solve %.1f ms host to host, %.1f ms on the device = %.1f ms per rhs (%.0f GB/s over %.1f GB of panels)  residual %.2e" % (
        N, a.facto, a.nrhs, dt * 1e3, dev * 1e3, dev * 1e3 / a.nrhs, nbytes / dev * 1e-9, nbytes * 1e-9, res), flush=True)
