#!/usr/bin/env python3
"""Host-only estimate of multi-GPU scaling from the per-rank schedules (no GPU needed).
Model: a launch of F flops with T tasks whose largest task has W multiply-adds takes
max(F / R_chip, (2W) / R_wg) + t0, ranks advance in lockstep per level."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastix_amd import symbolic as sy, dist as pd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R_chip, R_wg, t0 = 50e12, 50e12 / 400, 30e-6
n, cp, r, v = sy.laplacian_3d(N); perm, _ = sy.order_grid(N, N, N); s = sy.symbolic(n, cp, r, perm)
c4, b4 = s["cblk4"], s["blok4"]
base = None
for P in (1, 2, 4, 8):
    owner = pd.partition(c4, b4, P)
    per = [pd.plan_profile(c4, b4, owner if P > 1 else None, q) for q in range(P)]
    nl = len(per[0][0])
    tot = 0.0
    for l in range(nl):
        tl = 0.0
        for sf, sm, stn, pf in per:
            t = 0.0
            if stn[l] > 0: t += max(sf[l] / R_chip, 2 * sm[l] / R_wg) + t0
            if pf[l] > 0: t += pf[l] / (R_chip / 4) + 2 * t0
            tl = max(tl, t)
        tot += tl
    if base is None: base = tot
    print("P=%d levels=%d est time %.4f s speedup %.2f  rank flops share max %.3f" % (
        P, nl, tot, base / tot, max(p[0].sum() for p in per) / sum(p[0].sum() for p in per)))
