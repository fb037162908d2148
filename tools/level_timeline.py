#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of ONE factorization (last repetition traced): per decile of levels, the time
the panel stream (urgent update + diag + trsm) and the bulk stream are busy, and the wall time of the decile."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = "bulk" if "k_update<0>" in n else "urg" if "k_update<1>" in n else "trsm" if "k_trsm" in n else "diag" if "k_diag" in n else None
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
diag = [e for e in ev if e[2] == "diag"]
nlev = len(diag) // int(sys.argv[2]) if len(sys.argv) > 2 else len(diag)
diag = diag[-nlev:]
t0 = diag[0][0]
ev = [e for e in ev if e[0] >= t0 - 2_000_000]
bounds = [d[0] for d in diag] + [ev[-1][1]]
print("levels", nlev, "wall %.1f ms" % ((bounds[-1] - bounds[0]) / 1e6))
for i in range(10):
    a, b = bounds[i * nlev // 10], bounds[(i + 1) * nlev // 10]
    seg = [e for e in ev if a <= e[0] < b]
    busy = lambda ks: sum(e[1] - e[0] for e in seg if e[2] in ks) / 1e6
    print("levels %4d-%4d: wall %7.2f ms | panel stream busy %7.2f (urgent %6.2f diag %6.2f trsm %6.2f) | bulk busy %7.2f" % (
        i * nlev // 10, (i + 1) * nlev // 10, (b - a) / 1e6, busy(("urg", "diag", "trsm")), busy(("urg",)), busy(("diag",)), busy(("trsm",)), busy(("bulk",))))
