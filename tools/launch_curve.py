#!/usr/bin/env python3
"""Measured duration of a bulk update launch (k_update<0>, level-by-level schedule: what every rank of the multi-GPU
driver runs) as a function of its flops: runs tools/dev_bench.py at several sizes with options.verbose = 2
(PASTIX_AMD_RUN=0), parses the per-launch lines the engine prints ("bulk  l: tasks ... flops ... us"), and writes the
curve -- median rate per bin of log10(flops) -- as JSON.  tools/sim_scaling.py reads it instead of assuming a flat rate.
usage: launch_curve.py OUT.json [sizes ...]      (run on the GPU box)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
sizes = [int(x) for x in sys.argv[2:]] or [60, 100, 130, 160]
pat = re.compile(r"^bulk\s+(\d+): tasks\s+(\d+) .*?flops ([0-9.e+]+)\s+full.*?\s([0-9.]+) us\s+([0-9.]+) GF/s")
pts = []
for n in sizes:
    env = dict(os.environ, PASTIX_AMD_RUN="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev_bench.py"), "-n", str(n), "--reps", "2", "--verbose", "2"],
                       env=env, capture_output=True, text=True)
    rows = {}
    for line in r.stderr.splitlines():
        m = pat.match(line)
        if m:
            rows[int(m.group(1))] = (float(m.group(3)), int(m.group(2)), float(m.group(4)) * 1e-6)   # (last repetition wins)
    pts += [(n, l) + v for l, v in rows.items()]
import math
bins = {}
for n, l, fl, nt, t in pts:
    if fl <= 0 or t <= 0:
        continue
    b = round(math.log10(fl) * 4) / 4.0
    bins.setdefault(b, []).append(fl / t)
curve = []
for b in sorted(bins):
    v = sorted(bins[b])
    curve.append({"log10_flops": b, "launches": len(v), "median_TFLOPs": round(v[len(v) // 2] * 1e-12, 3),
                  "min_TFLOPs": round(v[0] * 1e-12, 3), "max_TFLOPs": round(v[-1] * 1e-12, 3)})
json.dump({"what": "bulk update launches of the level-by-level schedule (k_update<0> [+ k_update_small<0>]), duration by HIP events, "
                   "d LLt 3-D Laplacian, sizes %s: rate by launch size" % sizes,
           "curve": curve}, open(out, "w"), indent=1)
print(json.dumps(curve))
