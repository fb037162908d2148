#!/usr/bin/env python3
"""Per-rank efficiency of the level-stepped (distributed) engine on ONE GPU: a world-size-1 distributed plan, driven
level by level like pastix_amd.dist.factorize_levels; PASTIX_AMD_DIST_OVERLAP=0|1 selects one or two streams."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pastix_amd import dist as pd, fact_flops          # noqa: E402
from pastix_amd import symbolic as sy                   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, cp, r, v = sy.laplacian_3d(N)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
c4, b4 = s["cblk4"], s["blok4"]
owner = np.zeros(len(c4) - 1, dtype=np.int32)
eng = pd.GpuEngine(c4, b4, owner, 0, 0)
eng.fill_csc(1, n, cp, r, v, s["perm"])
nl = int(eng.level.max()) + 1
fl = fact_flops(c4, b4, 0)
for rep in range(3):
    eng.refill()
    eng.begin(1e-14)
    for l in range(nl):
        eng.update(l)
        eng.panels(l)
    st = eng.end()
    print("N=%d staged: fact %.4f s = %.1f GFLOP/s" % (N, st["fact_time"], fl / st["fact_time"] * 1e-9), flush=True)
eng.close()
