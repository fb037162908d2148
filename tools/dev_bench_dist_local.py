#!/usr/bin/env python3
"""Overhead check of the native multi-GPU driver on ONE GPU: W emulated ranks (loopback transport, one host thread
each) factorize the same problem the single-GPU driver does; the GPU does the same flops, so the difference is what
the fan-in buffers, the channel traffic (device-to-device here), the adds and the many small launches cost.
usage: dev_bench_dist_local.py GRID WORLD [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pastix_amd import Plan, fact_flops  # noqa: E402
from pastix_amd import dist as pd  # noqa: E402
from pastix_amd import symbolic as sy  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, cp, r, v = sy.laplacian_3d(N)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
c4, b4 = s["cblk4"], s["blok4"]
fl = fact_flops(c4, b4, 0)
crit = 1e-14
with Plan(c4, b4, 0) as p:
    p.fill_csc(1, n, cp, r, v, s["perm"])
    p.factorize(crit)
    best = 1e9
    for _ in range(reps):
        p.refill()
        best = min(best, p.factorize(crit)["fact_time"])
print("single GPU driver : %.4f s  %.1f TFLOP/s" % (best, fl / best * 1e-12), flush=True)
owner = pd.partition(c4, b4, W)
plans = [pd.DistPlan(c4, b4, owner, q, 0) for q in range(W)]
pd.attach_local(plans)
for q in plans:
    q.fill_csc(1, n, cp, r, v, s["perm"])
pd.factorize_local(plans, crit)
best = 1e9
for _ in range(reps):
    for q in plans:
        q.refill()
    t0 = time.time()
    st = pd.factorize_local(plans, crit)
    best = min(best, time.time() - t0)
ld = 2.0 * sum(q.diag_logsum() for q in plans)
cs = 2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1))
exact = float(np.log(6.0 - cs[:, None, None] - cs[None, :, None] - cs[None, None, :]).sum())
info = [q.info() for q in plans]
print("%d emulated ranks  : %.4f s  %.1f TFLOP/s  (per-rank device times %s)  log-det rel err %.1e; %d fan-in blocks, %.2f GB"
      % (W, best, fl / best * 1e-12, " ".join("%.3f" % x["fact_time"] for x in st), abs(ld - exact) / abs(exact),
         sum(i["nsend"] for i in info), sum(i["bytes_sent"] for i in info) * 1e-9), flush=True)
for q in plans:
    q.close()
