#!/usr/bin/env python3
"""Race screen for the update kernel's DMA/barrier pipeline and the two-stream driver: refactorize the same matrix
many times and require bitwise identical factors (tile ownership makes the summation order fixed, so any difference
is a synchronisation bug)."""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastix_amd import Plan                      # noqa: E402
from pastix_amd import symbolic as sy            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=60)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--facto", type=int, default=0)
a = ap.parse_args()
N = a.n
n, cp, r, v = sy.laplacian_3d(N, full=(a.facto == 2))
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
digests = set()
t0 = time.time()
with Plan(s["cblk4"], s["blok4"], a.facto) as p:
    p.fill_csc(0 if a.facto == 2 else 1, n, cp, r, v, s["perm"])
    for i in range(a.reps):
        p.refill()
        st = p.factorize(1e-14)
        L, U = p.download()
        h = hashlib.sha1(L.tobytes())
        if U is not None:
            h.update(U.tobytes())
        digests.add(h.hexdigest())
        if not np.isfinite(L).all():
            print("non-finite factor at rep", i)
            sys.exit(2)
print("n=%d^3 facto=%d: %d factorizations, %d distinct digests, %.1f s" % (N, a.facto, a.reps, len(digests), time.time() - t0))
sys.exit(0 if len(digests) == 1 else 1)
