#!/usr/bin/env python3
"""Per bulk launch of one factorization: update flops (host plan profile), duration (rocprofv3 kernel trace of
tools/dev_bench.py, last repetition), rate -- where the bulk stream loses against the in-situ whole-tile rate.
usage: launch_eff.py TRACE_DIR N"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pastix_amd import symbolic as sy
from pastix_amd import dist as pd

N = int(sys.argv[2])
n, cp, r, v = sy.laplacian_3d(N)
perm, invp = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm)
sf, sm, stn, pf, uf = pd.plan_profile(s["cblk4"], s["blok4"], None, 0)
bulk = sf - uf
slots = [i for i in range(len(sf)) if bulk[i] > 0]
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for row in csv.DictReader(open(f)):
    if "k_update<0>" in row["Kernel_Name"]:
        ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0)))
ev.sort()
ev = ev[-len(slots):]
assert len(ev) == len(slots), (len(ev), len(slots))
dur = np.array([e[1] - e[0] for e in ev]) * 1e-9
gap = np.array([0] + [ev[i][0] - ev[i - 1][1] for i in range(1, len(ev))]) * 1e-9
fl = bulk[slots]
tf = fl / dur * 1e-12
wg = np.array([e[2] for e in ev]) / 256
print("bulk launches %d, flops %.3e, busy %.2f ms (%.1f TF), gaps between them %.2f ms, span %.2f ms" % (
    len(ev), fl.sum(), dur.sum() * 1e3, fl.sum() / dur.sum() * 1e-12, gap.sum() * 1e3, (ev[-1][1] - ev[0][0]) * 1e-6))
print("%5s %8s %9s %8s %7s %8s %8s" % ("slot", "wgs", "GF", "ms", "TF", "gap_us", "maxtask%"))
for i, sl in enumerate(slots):
    if len(sys.argv) > 3:
        print("%5d %8d %9.1f %8.3f %7.1f %8.1f %8.1f" % (sl, wg[i], fl[i] * 1e-9, dur[i] * 1e3, tf[i], gap[i] * 1e6,
                                                       100 * 2 * sm[sl] / max(fl[i], 1)))
edges = [0, 20, 30, 40, 50, 55, 60, 65, 100]
for a, b in zip(edges[:-1], edges[1:]):
    m = (tf >= a) & (tf < b)
    print("rate %3d-%3d TF: %4d launches, %6.2f ms (%.1f%% of busy), %.1f%% of flops, mean wgs %.0f" % (
        a, b, m.sum(), dur[m].sum() * 1e3, 100 * dur[m].sum() / dur.sum(), 100 * fl[m].sum() / fl.sum(), wg[m].mean() if m.any() else 0))
for R in (62, 66):
    print("all launches at >= %d TF would give busy %.2f ms" % (R, np.minimum(dur, fl / (R * 1e12)).sum() * 1e3))
# rounds model: a launch of W workgroups on 512 slots
rounds = wg / 512.0
for a, b in [(0, 1), (1, 2), (2, 3), (3, 5), (5, 10), (10, 1e9)]:
    m = (rounds >= a) & (rounds < b)
    if m.any():
        print("rounds %4.0f-%4.0f: %4d launches, %6.2f ms, %.1f TF" % (a, min(b, 9999), m.sum(), dur[m].sum() * 1e3, fl[m].sum() / dur[m].sum() * 1e-12))
