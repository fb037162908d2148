import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(facto=None, prec="d"):
    """Fixture names; prec 'd' = real double, 'z' = complex double (names start with 'z'), None = all."""
    import glob
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))
    if facto:
        names = [n for n in names if n.split("_")[2] == facto]
    if prec == "d":
        names = [n for n in names if not n.startswith("z")]
    elif prec == "z":
        names = [n for n in names if n.startswith("z")]
    return names


@pytest.fixture(scope="session")
def golden():
    import fixture_io
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = fixture_io.load_npz(os.path.join(HERE, "golden", name + ".npz"))
        return cache[name]
    return get


def recut_mask(c4, group=128):
    """True where a packed panel entry is compared for LLt.  Everything, except the strict upper triangle of the
    diagonal blok of a cblk wider than 128 columns.  The strict upper triangle is not part of L and nothing reads it
    (potrf, trsm and updo use the lower triangle); the reference leaves by-products of its rectangular scatter there
    (add_contrib_local subtracts whole blok x blok rectangles, sopalin_compute.c:427-452) -- which the engine's tiles
    reproduce for every cblk of at most 128 columns, so those are compared in full.  A wider cblk is re-cut into column
    groups of 128 (api.cpp build_split): the blocks above the groups have no storage on the device (the engine hands the
    INPUT values back there), and inside a group the upper triangle also receives the earlier groups' update, which the
    reference's blocked potrf applies to the lower triangle only (SYRK "L", compute_diag.c:197-200)."""
    import numpy as np
    w = c4[:-1, 1] - c4[:-1, 0] + 1
    parts = []
    for k in range(len(w)):
        wk, sk = int(w[k]), int(c4[k, 3])
        m = np.ones((wk, sk), dtype=bool)            # [col][row]
        if wk > group:
            m[:, :wk] = np.triu(np.ones((wk, wk), dtype=bool))     # row >= col
        parts.append(m.ravel())
    return np.concatenate(parts)
