#!/usr/bin/env python3
"""Developer benchmark for BASELINE config 5: complex double LDLt on the 3-dof elasticity pattern."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastix_amd import Plan, fact_flops, COMPLEXDOUBLE
from pastix_amd import symbolic as sy
ap = argparse.ArgumentParser(); ap.add_argument("-n", type=int, default=24); ap.add_argument("--bs", type=int, default=128)
a = ap.parse_args()
n, cp, r, v, _ = sy.elasticity_3d(a.n)
perm, _ = sy.order_grid_dof(a.n, 3)
s = sy.symbolic(n, cp, r, perm, max_blocksize=a.bs)
c4, b4 = s["cblk4"], s["blok4"]
fl = fact_flops(c4, b4, 1, COMPLEXDOUBLE)
p = Plan(c4, b4, 1, floattype=COMPLEXDOUBLE)
p.fill_csc(1, n, cp, r, v, s["perm"])
for rep in range(3):
    p.refill()
    st = p.factorize(1e-12)
    print("z LDLt elasticity %d^3 x3 (n=%d, cblk %d): %.4f s = %.1f GFLOP/s (complex flops, %.1f%% of 78.6T); k_update %.1f GF/s; pivots %d" % (
        a.n, n, len(c4) - 1, st["fact_time"], fl / st["fact_time"] * 1e-9, fl / st["fact_time"] / 78.6e12 * 100,
        st["update_flops"] / max(st["update_time"], 1e-9) * 1e-9, st["nbpivot"]), flush=True)
# device solve (forward / D / backward on the split planes), host vector in -> host vector out
import scipy.sparse as sp
Al = sp.csc_matrix((v, r - 1, cp - 1), shape=(n, n))
A = Al + sp.tril(Al, -1).T
rng = np.random.default_rng(2)
xs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
b = A @ xs
pm = np.asarray(s["perm"])
for rep in range(3):
    bp = np.empty(n, dtype=np.complex128)
    bp[pm] = b
    t = time.time()
    x = p.solve(bp)[pm]
    dt = time.time() - t
    print("z solve %.2f ms (%.0f GB/s over the panels), residual %.2e" % (
        dt * 1e3, 2 * 16.0 * s["nnzl"] / dt * 1e-9, np.linalg.norm(A @ x - b) / np.linalg.norm(b)), flush=True)
