#!/usr/bin/env python3
"""Soak of the run schedule: factorize the same matrix `--reps` times and count the factorizations in which a wait inside
the run launch expired (the step is then redone on the level-by-level schedule: stats run_time == 0) or that failed.
Every `--check` steps the factors are downloaded and hashed: all hashes must agree (the run schedule is bitwise
deterministic).  One JSON line per configuration; exit code 1 if any factorization stopped, 2 on a wrong result.

  tools/soak_run.py -n 60 --facto 0 --reps 5000
  tools/soak_run.py -n 32 --dof 3 --facto 1 --complex --reps 2000      (z LDLt, elasticity pattern)
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastix_amd import Plan                      # noqa: E402
from pastix_amd import symbolic as sy            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=60)
ap.add_argument("--facto", type=int, default=0, help="0 LLt, 1 LDLt, 2 LU, 3 LDLh")
ap.add_argument("--complex", action="store_true", help="z arithmetic on the 3-dof elasticity pattern")
ap.add_argument("--reps", type=int, default=1000)
ap.add_argument("--check", type=int, default=0, help="hash the factors every this many steps (0: first and last only)")
ap.add_argument("--timeout", type=float, default=0.5, help="PASTIX_AMD_RUN_TIMEOUT for the soak (s): what a stop costs")
ap.add_argument("--tag", default="")
ap.add_argument("--force-run", action="store_true", help="options.run_schedule = 1: build the run whatever the size (200^3)")
a = ap.parse_args()
os.environ.setdefault("PASTIX_AMD_RUN_TIMEOUT", str(a.timeout))
os.environ.setdefault("PASTIX_AMD_LAUNCH_EVENTS", "1")

N = a.n
if a.complex:
    assert a.facto == 1, "the soak's complex case is z LDLt on the elasticity pattern (configs[4])"
    n, cp, r, v, _ = sy.elasticity_3d(N)
    perm, _ = sy.order_grid_dof(N, 3)
    sym, ft = 1, 3
else:
    n, cp, r, v = sy.laplacian_3d(N, full=(a.facto == 2))
    perm, _ = sy.order_grid(N, N, N)
    sym, ft = (0 if a.facto == 2 else 1), 1
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
stops = errors = 0
digests = {}
times = []
t0 = time.time()
with Plan(s["cblk4"], s["blok4"], a.facto, floattype=ft, run_schedule=1 if a.force_run else 0) as p:
    p.fill_csc(sym, n, cp, r, v, s["perm"])
    st = p.factorize(1e-14)
    has_run = st["run_time"] > 0
    for i in range(a.reps):
        p.refill()
        try:
            st = p.factorize(1e-14)
        except Exception as e:                   # noqa: BLE001
            errors += 1
            print("step %d failed: %s" % (i, e), flush=True)
            continue
        if has_run and st["run_time"] == 0:
            stops += 1
            print("step %d: redone on the level schedule (%.1f s into the soak)" % (i, time.time() - t0), flush=True)
        else:
            times.append(st["fact_time"])
        if (a.check and i % a.check == 0) or i in (0, a.reps - 1):
            L, U = p.download()
            h = hashlib.sha1(L.tobytes())
            if U is not None:
                h.update(U.tobytes())
            digests[h.hexdigest()] = digests.get(h.hexdigest(), 0) + 1
            del L, U
tt = np.array(times) if times else np.zeros(1)
out = {"n": N, "facto": a.facto, "complex": bool(a.complex), "reps": a.reps, "run": bool(has_run), "stops": stops, "errors": errors,
       "distinct_digests": len(digests), "median_ms": float(np.median(tt) * 1e3), "p99_ms": float(np.percentile(tt, 99) * 1e3),
       "max_ms": float(tt.max() * 1e3), "wall_s": round(time.time() - t0, 1), "tag": a.tag,
       "env": {k: v for k, v in os.environ.items() if k.startswith("PASTIX_AMD_")}}
print(json.dumps(out), flush=True)
sys.exit(2 if len(digests) > 1 or errors else 1 if stops else 0)
