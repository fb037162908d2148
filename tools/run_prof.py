#!/usr/bin/env python3
"""Reads a PASTIX_AMD_DEV=run_prof=<file> dump (api.cpp): clock stamps (100 MHz) of every ticket of the run launch and of the
resident diagonal tasks.  Prints slot-time by category (waiting / running) and a per-level chain timeline."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
nu, nd, _, L0 = [int(x) for x in raw[:4]]
n = nu + nd
st = raw[4:4 + 4 * n].reshape(n, 4)
cl = raw[4 + 4 * n:4 + 4 * n + nu]
cat = cl & 255
lvl = cl >> 8
U = st[:nu]; D = st[nu:]
t0 = U[:, 0].min(); t1 = max(U[:, 2].max(), D[:, 2].max() if nd else 0)
wall = (t1 - t0) * 1e-8
print("run: %d tickets, %d diag tasks; wall %.3f ms; first level %d" % (nu, nd, wall * 1e3, L0))
names = ["A (urgent, targets of the level)", "B.next (targets of next level)", "B.rest", "T (panel solve, 128 rows)"]
tot_slot = 0
for c in range(4):
    m = cat == c
    if not m.any(): continue
    w = (U[m, 1] - U[m, 0]).sum() * 1e-8; r = (U[m, 2] - U[m, 1]).sum() * 1e-8
    tot_slot += w + r
    print("  %-34s %8d: waiting %9.3f ms  running %9.3f ms (slot-time; mean wait %.1f us, mean run %.1f us)" % (
        names[c], m.sum(), w * 1e3, r * 1e3, w / m.sum() * 1e6, r / m.sum() * 1e6))
print("  ticket slot-time total %.3f ms = %.1f slots busy on average (of 512)" % (tot_slot * 1e3, tot_slot / wall))
if nd:
    print("  diag tasks: waiting %.3f ms running %.3f ms (mean run %.1f us; last 20: %.1f us)" % (
        (D[:, 1] - D[:, 0]).sum() * 1e-5, (D[:, 2] - D[:, 1]).sum() * 1e-5, (D[:, 2] - D[:, 1]).mean() * 1e-2, (D[-20:, 2] - D[-20:, 1]).mean() * 1e-2))
mt = cat == 3
if mt.any():
    top = mt & (lvl >= lvl.max() - 20)
    print("  T at the top 20 levels: mean run %.1f us, mean wait %.1f us" % ((U[top, 2] - U[top, 1]).mean() * 1e-2, (U[top, 1] - U[top, 0]).mean() * 1e-2))
# per CU: time covered by tickets (two workgroups fit a CU)
hw = U[:, 3]
cu = ((hw >> 32) << 16) | (hw & 0xff00)       # xcc | se, sh, cu bits of HW_ID
cus = np.unique(cu)
occ = 0.0
for c in cus[:64]:
    m = cu == c
    occ += (U[m, 2] - U[m, 0]).sum() * 1e-8
print("  %d distinct CU ids; on the first %d of them tickets cover %.1f %% of 2 slots x wall" % (len(cus), min(64, len(cus)), 100 * occ / (min(64, len(cus)) * 2 * wall)))
order = np.argsort(U[:, 0])
gaps = []
for c in cus[:16]:
    m = np.where(cu == c)[0]
    ev = sorted([(U[i, 0], 1) for i in m] + [(U[i, 2], -1) for i in m])
    lvl2 = 0; last = t0; idle1 = 0; idle2 = 0
    for tme, d in ev:
        if lvl2 == 0: idle2 += 2 * (tme - last)
        elif lvl2 == 1: idle1 += (tme - last)
        lvl2 += d; last = tme
    gaps.append((idle1 + idle2) * 1e-8 / (2 * wall))
print("  idle share of the two slots on 16 CUs: mean %.3f min %.3f max %.3f" % (np.mean(gaps), np.min(gaps), np.max(gaps)))
print("level: A first drawn / last ready / last done | T last done | period (us)")
rows = []
for s in range(L0, int(lvl.max()) + 1):
    m = (cat == 0) & (lvl == s)
    if not m.any(): continue
    t = (cat == 3) & (lvl == s)
    rows.append((s, (U[m, 0].min() - t0) * 1e-2, (U[m, 1].max() - t0) * 1e-2, (U[m, 2].max() - t0) * 1e-2, int(m.sum()),
                 (U[t, 2].max() - t0) * 1e-2 if t.any() else 0.0, (U[t, 1] - U[t, 0]).mean() * 1e-2 if t.any() else 0.0))
step = max(1, len(rows) // 40)
for i, (s, a0, ar, ad, k, td, tw) in enumerate(rows):
    if i % step == 0 or i >= len(rows) - 3:
        per = (ad - rows[i - 1][3]) if i > 0 else 0
        print("  %4d: n=%4d drawn %10.1f ready %10.1f done %10.1f | T done %10.1f (mean wait %6.1f) | A spin %7.1f period %7.1f" % (
            s, k, a0, ar, ad, td, tw, ar - a0, per))
if len(sys.argv) > 2:   # chain detail of the last K levels: every stamp relative to the level's first event
    K = int(sys.argv[2])
    print("chain detail (us from run start): D start/end | T ready(min) done(min,max) run(mean) | A ready(min,max) done(min,max)")
    Dl = D[-K:] if nd else []
    for i, s in enumerate(range(int(lvl.max()) - K + 1, int(lvl.max()) + 1)):
        a = (cat == 0) & (lvl == s); t = (cat == 3) & (lvl == s)
        f = lambda x: (x - t0) * 1e-2
        ds = " D %9.1f %9.1f (%5.1f)" % (f(Dl[i][1]), f(Dl[i][2]), (Dl[i][2] - Dl[i][1]) * 1e-2) if nd else ""
        ts = " | T n=%3d ready %9.1f done %9.1f..%9.1f run %5.1f" % (t.sum(), f(U[t, 1].min()), f(U[t, 2].min()), f(U[t, 2].max()), (U[t, 2] - U[t, 1]).mean() * 1e-2) if t.any() else ""
        as_ = " | A n=%3d ready %9.1f..%9.1f done %9.1f..%9.1f" % (a.sum(), f(U[a, 1].min()), f(U[a, 1].max()), f(U[a, 2].min()), f(U[a, 2].max())) if a.any() else ""
        print("  %4d:%s%s%s" % (s, ds, ts, as_))
if len(sys.argv) > 4:   # every ticket that became ready in [a, b] us
    a, b = float(sys.argv[3]), float(sys.argv[4])
    r = (U[:, 1] - t0) * 1e-2
    sel = np.where((r >= a) & (r <= b))[0]
    for i in sel[np.argsort(r[sel])]:
        print("   ticket %7d cat %d lvl %4d drawn %10.1f ready %10.1f done %10.1f (run %5.1f) cu %x" % (i, cat[i], lvl[i], (U[i, 0] - t0) * 1e-2, r[i], (U[i, 2] - t0) * 1e-2, (U[i, 2] - U[i, 1]) * 1e-2, cu[i]))
