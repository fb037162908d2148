#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace csv: per kernel, time by launch-order decile (levels go bottom-up)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    key = "update" if "k_update" in n else "trsm" if "k_trsm" in n else "diag" if "k_diag" in n else None
    if key: rows[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in rows.items():
    v.sort()
    v = v[len(v) // 2:] if len(v) > 800 else v          # last repetition if two were traced
    d = [x[1] for x in v]
    n = len(d)
    print(k, "launches", n, "total %.1f ms" % (sum(d) / 1e6))
    for i in range(10):
        seg = d[i * n // 10:(i + 1) * n // 10]
        print("   launches %4d-%4d: %8.2f ms  avg %8.1f us  max %8.1f us" % (i * n // 10, (i + 1) * n // 10, sum(seg) / 1e6, sum(seg) / max(len(seg), 1) / 1e3, max(seg) / 1e3))
