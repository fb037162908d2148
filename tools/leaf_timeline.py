#!/usr/bin/env python3
"""First levels of one factorization from a rocprofv3 kernel trace (tools/dev_bench.py, last repetition): every kernel
with start, duration and workgroups -- what the leaf chains are bound by.  usage: leaf_timeline.py TRACE_DIR [nkernels]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 90
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = ("small_u" if "k_update_small<1>" in n else "small" if "k_update_small" in n else "bulk" if "k_update<0>" in n else
         "urgent" if "k_update<1>" in n else "trsm" if "k_trsm" in n else "diag" if "k_diag" in n else None)
    if k:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                   int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)))
ev.sort()
diag = [e for e in ev if e[2] == "diag"]
nl = len(diag) // 2
t0 = diag[-nl][0] - 200_000
ev = [e for e in ev if e[0] >= t0]
tot = {}
for e in ev[:nk]:
    print("%8s start %9.3f ms dur %8.3f ms wgs %6d" % (e[2], (e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3] // max(e[4], 1)))
lim = ev[min(nk, len(ev) - 1)][0]
for e in ev:
    if e[0] < lim: tot[e[2]] = tot.get(e[2], 0) + (e[1] - e[0]) / 1e6
print("busy ms in this window (%.2f ms wall):" % ((lim - t0) / 1e6), {k: round(v, 2) for k, v in tot.items()})
