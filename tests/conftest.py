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
