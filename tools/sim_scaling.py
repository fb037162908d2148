#!/usr/bin/env python3
"""Host-only estimate of multi-GPU scaling from the per-rank schedules (no GPU needed).
Model: a launch of F flops with T tasks whose largest task has W multiply-adds takes
max(F / R_chip, (2W) / R_wg) + t0, ranks advance in lockstep per level; a level with fan-in traffic adds
t_msg + (largest per-rank bytes sent or received in that level) / BW_rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastix_amd import symbolic as sy, dist as pd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R_chip, R_wg, t0 = 60e12, 60e12 / 400, 30e-6
t_msg, BW_rank = 80e-6, 150e9      # per-level exchange latency; per-rank aggregate xGMI rate assumed (of 7 x 153 GB/s peak)
n, cp, r, v = sy.laplacian_3d(N); perm, _ = sy.order_grid(N, N, N); s = sy.symbolic(n, cp, r, perm)
c4, b4 = s["cblk4"], s["blok4"]
base = None
for P in (1, 2, 4, 8):
    owner = pd.partition(c4, b4, P)
    per = [pd.plan_profile(c4, b4, owner if P > 1 else None, q) for q in range(P)]
    nl = len(per[0][0])
    tot = 0.0
    comm = np.zeros((nl, P, 2))
    if P > 1:
        level = pd.levels_of(c4, b4)
        mask = pd.fanin_touched(c4, b4, owner)
        w = (c4[:-1, 1] - c4[:-1, 0] + 1)
        cb = np.repeat(np.arange(len(c4) - 1), np.diff(c4[:, 2]))
        h = b4[:, 1] - b4[:, 0] + 1
        for q in range(P):
            byt = 8.0 * h * w[cb] * ((mask >> np.uint64(q)) & np.uint64(1))
            np.add.at(comm[:, q, 0], level[cb], byt)                       # sent by q at the target's level
            np.add.at(comm[:, :, 1], (level[cb], owner[cb]), byt)          # received by the owner
    tcomm = 0.0
    for l in range(nl):
        tl = 0.0
        if comm[l].max() > 0:
            tc = t_msg + comm[l].max() / BW_rank
            tot += tc
            tcomm += tc
        for sf, sm, stn, pf in per:
            t = 0.0
            if stn[l] > 0: t += max(sf[l] / R_chip, 2 * sm[l] / R_wg) + t0
            if pf[l] > 0: t += pf[l] / (R_chip / 4) + 2 * t0
            tl = max(tl, t)
        tot += tl
    if base is None: base = tot
    print("P=%d levels=%d est time %.4f s (fan-in %.4f s, %.2f GB sent by the busiest rank) speedup %.2f  rank flops share max %.3f" % (
        P, nl, tot, tcomm, comm[:, :, 0].sum(axis=0).max() * 1e-9, base / tot,
        max(p[0].sum() for p in per) / sum(p[0].sum() for p in per)), flush=True)
