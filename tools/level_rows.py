#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of ONE factorization (the last one traced): one row per level of the level-by-level
part -- when its diagonal kernel starts, how long the urgent updates, the diagonal and panel-solve kernels and the bulk
launches take, and the idle time of the panel stream before the diagonal kernel.  usage: level_rows.py DIR [NLEVELS]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = ("run" if "k_run_update" in n else "rund" if "k_run_diag" in n else "bulk" if "k_update<0" in n else "urg" if "k_update<1" in n
         else "trsm" if "k_trsm" in n else "diag" if "k_diag" in n else "fill" if ("k_fill" in n or "k_scatter" in n) else None)
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
fills = [e for e in ev if e[2] == "fill"]
t_last_fill = fills[-1][1] if fills else ev[0][0]
ev = [e for e in ev if e[0] >= t_last_fill and e[2] != "fill"]
t0 = ev[0][0]
diag = [e for e in ev if e[2] == "diag"]
print("level-by-level part: %d diagonal launches; first kernel at 0, run launch at %s" % (
    len(diag), ", ".join("%.2f ms" % ((e[0] - t0) / 1e6) for e in ev if e[2] == "run") or "-"))
print("level: diag start(ms) | urgent before it (us, n) | diag us | trsm us (n) | bulk launches overlapping [start of diag, next diag): busy us | level wall us")
for i, d in enumerate(diag[:nshow]):
    nxt = diag[i + 1][0] if i + 1 < len(diag) else max(e[1] for e in ev if e[2] in ("trsm", "diag"))
    prev = diag[i - 1][0] if i else t0
    urg = [e for e in ev if e[2] == "urg" and prev <= e[0] < d[0]]
    tr = [e for e in ev if e[2] == "trsm" and d[0] <= e[0] < nxt]
    bk = [e for e in ev if e[2] == "bulk" and e[1] > d[0] and e[0] < nxt]
    busy = sum(min(e[1], nxt) - max(e[0], d[0]) for e in bk)
    print("  %3d: %8.3f | %7.1f (%d) | %7.1f | %7.1f (%d) | %8.1f | %8.1f" % (
        i, (d[0] - t0) / 1e6, sum(e[1] - e[0] for e in urg) / 1e3, len(urg), (d[1] - d[0]) / 1e3,
        sum(e[1] - e[0] for e in tr) / 1e3, len(tr), busy / 1e3, (nxt - d[0]) / 1e3))
