#!/usr/bin/env python3
"""Developer A/B: the run schedule (one dependency-driven launch for the thin levels) against the level-by-level
schedule on the SAME plan: factors must be bitwise equal; prints both times."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from pastix_amd import Plan, fact_flops  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, nargs="+", default=[20])
ap.add_argument("--bs", type=int, default=128)
ap.add_argument("--maxc", type=int, default=0)
ap.add_argument("--tw", type=int, default=0)
ap.add_argument("--dw", type=int, default=0)
ap.add_argument("--look", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--nocheck", action="store_true")
ap.add_argument("--verbose", type=int, default=0)
ap.add_argument("--force-run", action="store_true", help="options.run_schedule = 1: build the run whatever the size")
a = ap.parse_args()
for N in a.n:
    n, cp, r, v = sy.laplacian_3d(N)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=a.bs)
    c4, b4 = s["cblk4"], s["blok4"]
    fl = fact_flops(c4, b4, 0)
    import time as _t
    _t0 = _t.time()
    p = Plan(c4, b4, 0, run_max_cblks=a.maxc, run_t_workers=a.tw, run_d_workers=a.dw, lookahead=a.look, verbose=a.verbose, run_schedule=1 if a.force_run else 0)
    print("N=%d plan created in %.2f s (maxc %d, force_run %s)" % (N, _t.time() - _t0, a.maxc, a.force_run), flush=True)
    res = {}
    for mode in ("0", "1"):
        os.environ["PASTIX_AMD_RUN"] = mode
        best = 1e9
        for rep in range(a.reps):
            p.fill_csc(1, n, cp, r, v, s["perm"])
            st = p.factorize(1e-14)
            best = min(best, st["fact_time"])
        L = None if a.nocheck else p.download()[0]
        res[mode] = (best, L, st)
        print("N=%d run=%s: %.3f ms = %.1f GFLOP/s  launches %d update_sum %.3f ms nbpivot %d" % (
            N, mode, best * 1e3, fl / best * 1e-9, st["nupdate_launches"], st["update_time_sum"] * 1e3, st["nbpivot"]), flush=True)
    if not a.nocheck:
        d = np.abs(res["0"][1] - res["1"][1]).max()
        print("N=%d max|L_run - L_levels| = %.3e  (bitwise equal: %s)  speedup %.3f" % (
            N, d, bool(np.array_equal(res["0"][1], res["1"][1])), res["0"][0] / res["1"][0]), flush=True)
    else:
        print("N=%d speedup %.3f" % (N, res["0"][0] / res["1"][0]), flush=True)
    p.close()
