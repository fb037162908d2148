#!/usr/bin/env python3
"""Developer benchmark: layout -> plan -> device fill -> factorize, prints stats (not bench.py)."""
import argparse
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from pastix_amd import Plan, fact_flops  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=60)
ap.add_argument("--leaf", type=int, default=8)
ap.add_argument("--amalg", type=int, default=5)
ap.add_argument("--bs", type=int, default=128)
ap.add_argument("--look", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--check", action="store_true")
ap.add_argument("--verbose", type=int, default=0)
ap.add_argument("--f32", action="store_true", help="single precision engine (kernels_f32.hip)")
a = ap.parse_args()
N = a.n
t = time.time()
n, cp, r, v = sy.laplacian_3d(N)
perm, invp = sy.order_grid(N, N, N, leaf=a.leaf)
s = sy.symbolic(n, cp, r, perm, max_blocksize=a.bs, amalgamation_pct=a.amalg)
c4, b4 = s["cblk4"], s["blok4"]
fl = fact_flops(c4, b4, 0)
print("N=%d cblk=%d blok=%d nnzl=%.3e flops=%.4e  symbolic %.1fs" % (N, len(c4) - 1, len(b4), s["nnzl"], fl, time.time() - t), flush=True)
t = time.time()
p = Plan(c4, b4, 0, lookahead=a.look, verbose=a.verbose, floattype=0 if a.f32 else 1)
st = p.stats()
print("plan %.1fs: levels=%d tasks=%d pieces=%d update_flops=%.4e (%.3f of total), in full pieces %.3f" % (
    time.time() - t, st["nlevels"], st["ntasks"], st["npieces"], st["update_flops"], st["update_flops"] / fl,
    st["full_flops"] / max(st["update_flops"], 1)), flush=True)
crit = 1e-14
for rep in range(a.reps):
    t = time.time()
    p.fill_csc(1, n, cp, r, v, s["perm"])
    tf = time.time() - t
    st = p.factorize(crit)
    print("rep %d: fill %.2fs fact %.4fs = %.1f GFLOP/s (%.1f%% of 78.6T) | update kernels %.4fs (%.1f GF/s on update flops) launches=%d nbpivot=%d" % (
        rep, tf, st["fact_time"], fl / st["fact_time"] * 1e-9, fl / st["fact_time"] / 78.6e12 * 100,
        st["update_time"], st["update_flops"] / max(st["update_time"], 1e-9) * 1e-9, st["nupdate_launches"], st["nbpivot"]), "sum %.4fs" % st["update_time_sum"], flush=True)
if a.check:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib
    L1, _ = p.download()
    L0, _ = oracle_lib.fill(0, 1, n, cp, r, v, s["perm"], c4, b4)
    Lo, _, nb = oracle_lib.sopalin(0, c4, b4, L0, None, crit)
    print("max|L_gpu - L_oracle| / max|L| = %.3e" % (np.abs(L1.astype(np.float64) - Lo).max() / np.abs(Lo).max()))
