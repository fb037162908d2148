#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (all streams merged), last repetition of
tools/dev_bench.py: histogram of the gaps in front of every kernel kind.  usage: gap_stats.py TRACE_DIR"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = ("small" if "k_update_small" in n else "bulk" if "k_update<0>" in n else "urgent" if "k_update<1>" in n else
         "trsm" if "k_trsm" in n else "diag" if "k_diag" in n else None)
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
diag = [e for e in ev if e[2] == "diag"]
t0 = diag[-(len(diag) // 2)][0] - 100_000
ev = [e for e in ev if e[0] >= t0]
end = 0
gaps = {}
for s, e, k in ev:
    if end and s > end: gaps.setdefault(k, []).append((s - end) / 1e3)
    elif end: gaps.setdefault(k, []).append(0.0)
    end = max(end, e)
tot = (ev[-1][1] - ev[0][0]) / 1e6
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, k in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms, at least one kernel running %.2f ms, idle %.2f ms" % (tot, busy / 1e6, tot - busy / 1e6))
for k, g in gaps.items():
    g = np.array(g)
    print("%7s: %4d starts, idle in front: mean %.1f us, median %.1f, p90 %.1f, sum %.2f ms" % (k, len(g), g.mean(), np.median(g), np.percentile(g, 90), g.sum() / 1e3))
