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
