import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def appendix_b():
    with open(os.path.join(GOLDEN, "appendix_b.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_m4():
    with open(os.path.join(GOLDEN, "oracle_m4.json")) as fh:
        d = json.load(fh)
    for k in ("val", "f", "u"):
        d[k] = np.array([float.fromhex(v) for v in d[k]])
    d["rowptr"] = np.array(d["rowptr"], np.int32)
    d["colidx"] = np.array(d["colidx"], np.int32)
    return d


@pytest.fixture(scope="session")
def golden_m32():
    return dict(np.load(os.path.join(GOLDEN, "oracle_m32.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def spk():
    """The product package; building it is __graft_entry__.build()'s job, but a
    missing library is built here so the CPU suite is self-contained."""
    so = os.path.join(ROOT, "saddle_point_petsc_amd", "libspk.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "saddle_point_petsc_amd", "csrc"), "-s"])
    import saddle_point_petsc_amd as S
    return S


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (d if d > 0 else 1.0)
