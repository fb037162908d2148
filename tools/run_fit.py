#!/usr/bin/env python3
"""Reads a PASTIX_AMD_DEV=run_prof=<file> dump (api.cpp) and fits what an update ticket of the run costs:
duration ~ a + b x (chunks on the branch-free loop) + c x (masked chunks) + d x (masked chunks x busiest wave's share) + e x pieces
+ f x (tile entries).  Prints the slot-time of the run by ticket class and the fitted unit costs."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
nu, nd, _, L0 = [int(x) for x in raw[:4]]
n = nu + nd
st = raw[4:4 + 4 * n].reshape(n, 4)
cl = raw[4 + 4 * n:4 + 4 * n + nu]
ft = raw[4 + 4 * n + nu:4 + 4 * n + nu + 8 * nu].reshape(nu, 8).astype(float)
cat = cl & 255
U = st[:nu]
run = (U[:, 2] - U[:, 1]) * 1e-2          # us
wait = (U[:, 1] - U[:, 0]) * 1e-2
upd = cat != 3
wall = (st[:, 2].max() - U[:, 0].min()) * 1e-2
print("wall %.1f us; update tickets %d, panel-solve tickets %d, diagonal tasks %d" % (wall, upd.sum(), (~upd).sum(), nd))
tot = run.sum() + wait.sum()
print("slot-time: update run %.1f %%, panel-solve run %.1f %%, waiting %.1f %%; of 512 slots x wall: %.1f %% covered" % (
    100 * run[upd].sum() / tot, 100 * run[~upd].sum() / tot, 100 * wait.sum() / tot, 100 * tot / (512 * wall)))
if nd:
    D = st[nu:]
    print("diagonal tickets: run %.1f us mean, %.2f %% of slot-time" % (((D[:, 2] - D[:, 1]) * 1e-2).mean(), 100 * ((D[:, 2] - D[:, 1]) * 1e-2).sum() / (512 * wall)))
F = ft[upd]
r = run[upd]
flops, cf, c1, c2, w1, w2, npc, ent = [F[:, i] for i in range(8)]
cm = c1 + c2
wm = (w1 + w2) / 8.0
X = np.stack([np.ones_like(r), cf, cm, wm, npc, ent / 16384.0], axis=1)
coef, *_ = np.linalg.lstsq(X, r, rcond=None)
names = ["per ticket", "per branch-free chunk", "per masked chunk (floor)", "per masked chunk x busiest share", "per piece", "per full tile of C"]
pred = X @ coef
print("fit (us): " + ", ".join("%s %.3f" % (nm, c) for nm, c in zip(names, coef)), " | rms residual %.1f us of mean %.1f" % (np.sqrt(((pred - r) ** 2).mean()), r.mean()))
parts = X * coef
print("share of the update tickets' run time by term: " + ", ".join("%s %.1f %%" % (nm, 100 * parts[:, i].sum() / pred.sum()) for i, nm in enumerate(names)))
print("flops: %.3e in update tickets; branch-free chunks %.3e, masked chunks %.3e (fill of masked chunks %.2f)" % (
    flops.sum(), cf.sum(), cm.sum(), (flops.sum() - cf.sum() * 524288) / max(cm.sum() * 524288, 1)))
# classes
allfull = (cm == 0)
for nm, m in [("whole-tile tasks", allfull), ("tasks with masked chunks", ~allfull)]:
    print("  %-26s %8d tickets, %5.1f %% of run time, %5.1f %% of flops, %.1f GFLOP/s per slot" % (
        nm, m.sum(), 100 * r[m].sum() / r.sum(), 100 * flops[m].sum() / flops.sum(), flops[m].sum() / r[m].sum() * 1e-3))
