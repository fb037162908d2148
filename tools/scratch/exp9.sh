#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export OPENBLAS_NUM_THREADS=1 REF_ORDER_CONTIG=1
for g in -1 0; do for k in 1 0; do echo "gather=$g onek=$k RUN=1 60 lu"; PASTIX_AMD_DEV="gather=$g,onek=$k" PASTIX_AMD_RUN=1 timeout 300 oracle/_ref/ref_harness_d_ob_amd cmp rlap3d 60 lu 32 /dev/null 2>/dev/null | grep '"cmp"' | cut -c1-180; done; done
python - <<'PY'
import os, sys, hashlib
sys.path.insert(0, '.')
import numpy as np
from pastix_amd import Plan
from pastix_amd import symbolic as sy
for N in (40, 60):
    n, cp, r, v = sy.laplacian_3d(N, full=True)
    perm, _ = sy.order_grid(N, N, N)
    s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
    for g in (-1, 0, 2):
        with Plan(s["cblk4"], s["blok4"], 2, gather_min=g) as p:
            out = {}
            for mode in ("0", "1"):
                os.environ["PASTIX_AMD_RUN"] = mode
                p.fill_csc(0, n, cp, r, v, s["perm"])
                st = p.factorize(1e-14)
                L, U = p.download()
                out[mode] = (L, U, st["run_tickets"])
            print("own layout LU", N, "gather_min", g, "run tickets", out["1"][2], "bitwise L", np.array_equal(out["0"][0], out["1"][0]), "U", np.array_equal(out["0"][1], out["1"][1]),
                  "maxdiff", float(np.abs(out["0"][0] - out["1"][0]).max()), flush=True)
PY
