import os, sys
sys.path.insert(0, '.')
import numpy as np
from pastix_amd import Plan
from pastix_amd import symbolic as sy
import ctypes
from pastix_amd import _lib
N = 48
n, cp, r, v = sy.laplacian_3d(N, full=True)
perm, _ = sy.order_grid(N, N, N)
s = sy.symbolic(n, cp, r, perm, max_blocksize=128)
c4, b4 = s["cblk4"], s["blok4"]
w = c4[:-1, 1] - c4[:-1, 0] + 1
st = c4[:-1, 3]
off = np.concatenate([[0], np.cumsum(w * st)])
with Plan(c4, b4, 2) as p:
    lvl = np.zeros(len(w), dtype=np.int32)
    poff = np.zeros(len(w) + 1, dtype=np.int64)
    role = np.zeros(len(w), dtype=np.int8)
    _lib.lib().pastix_amd_plan_layout_info(p._h, _lib.ptr(poff), _lib.ptr(lvl), _lib.ptr(role))
    os.environ["PASTIX_AMD_RUN"] = "0"
    p.fill_csc(0, n, cp, r, v, s["perm"]); p.factorize(1e-14)
    L0, U0 = p.download()
    os.environ["PASTIX_AMD_RUN"] = "1"
    found = 0
    for rep in range(400):
        p.refill(); p.factorize(1e-14)
        L1, U1 = p.download()
        if np.array_equal(L0, L1) and np.array_equal(U0, U1):
            continue
        found += 1
        # first differing cblk in level order
        bad = []
        for k in range(len(w)):
            a, b = off[k], off[k + 1]
            dl = L0[a:b] != L1[a:b]; du = U0[a:b] != U1[a:b]
            if dl.any() or du.any():
                bad.append((int(lvl[k]), k, int(dl.sum()), int(du.sum())))
        bad.sort()
        l0, k, ndl, ndu = bad[0]
        print("rep %d: %d cblks differ; first (lowest level %d): cblk %d width %d stride %d  L diffs %d  U diffs %d; same-level others: %s" % (
            rep, len(bad), l0, k, w[k], st[k], ndl, ndu, [b for b in bad[1:6] if b[0] == l0]), flush=True)
        a, b = off[k], off[k + 1]
        DL = (L0[a:b] != L1[a:b]).reshape(w[k], st[k]).T      # [row][col]
        DU = (U0[a:b] != U1[a:b]).reshape(w[k], st[k]).T
        for name, D in (("L", DL), ("U", DU)):
            rows, cols = np.nonzero(D)
            if len(rows) == 0: continue
            tiles = sorted(set(zip((rows // 16).tolist(), (cols // 16).tolist())))
            print("   %s arena: rows %d..%d cols %d..%d; 16x16 tiles (row band, col band): %s" % (name, rows.min(), rows.max(), cols.min(), cols.max(), tiles[:40]), flush=True)
            # detail of the first tile
            tr, tc = tiles[0]
            sub = D[tr*16:tr*16+16, tc*16:tc*16+16]
            print("   first tile pattern (rows x cols):"); 
            for rr in range(sub.shape[0]): print("     " + "".join("X" if x else "." for x in sub[rr]))
            M0 = L0 if name == "L" else U0; M1 = L1 if name == "L" else U1
            A0 = M0[a:b].reshape(w[k], st[k]).T; A1 = M1[a:b].reshape(w[k], st[k]).T
            rr, cc = rows[0], cols[0]
            print("   e.g. (%d,%d): level %.17g  run %.17g" % (rr, cc, A0[rr, cc], A1[rr, cc]))
        if found >= 3: break
    print("found", found)
