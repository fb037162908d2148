#!/usr/bin/env python3
"""What a drop-in caller pays: the one-shot entry point pastix_amd_d_po_sopalin (= D_po_sopalin_thread: host panels in, host
panels out) called three times on the same layout with fresh values -- the first call analyses the layout and allocates,
the later ones reuse the cached plan.  Prints plan / host->device / factorization / device->host / total per call as JSON.

  tools/one_shot_timing.py 100"""
import json
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastix_amd import Plan, fact_flops                 # noqa: E402
from pastix_amd import symbolic as sy                   # noqa: E402
from pastix_amd.solver import sopalin_tabs              # noqa: E402
from pastix_amd import _lib                             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, cp, r, v = sy.laplacian_3d(N)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
c4, b4 = s["cblk4"], s["blok4"]
with Plan(c4, b4, 0) as p:                               # the input panels, as CoefMatrix_Init would leave them on the host
    p.fill_csc(1, n, cp, r, v, s["perm"])
    L0 = p.download()[0]
w = c4[:-1, 1] - c4[:-1, 0] + 1
sz = w * c4[:-1, 3]
off = np.concatenate([[0], np.cumsum(sz)])
out = {"grid": N, "panel_bytes": int(8 * off[-1]), "fact_flops": fact_flops(c4, b4, 0), "calls": []}
for call in range(3):
    tabs = [L0[off[k]:off[k + 1]].copy() for k in range(len(w))]      # (separate host buffers, like cblktab[k].coeftab)
    t0 = time.time()
    st = sopalin_tabs(0, c4, b4, tabs, critere=1e-14)
    wall = time.time() - t0
    out["calls"].append({k: round(st[k], 4) for k in ("plan_time", "h2d_time", "fact_time", "d2h_time", "total_time")} | {"wall_s": round(wall, 4)})
    del tabs
_lib.lib().pastix_amd_release_cached_plan()
c = out["calls"][-1]
out["later_call_budget_s"] = round(c["fact_time"] + 2 * out["panel_bytes"] / 50e9 + 0.05, 4)
print(json.dumps(out))
