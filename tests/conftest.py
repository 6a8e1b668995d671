import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gorio():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("go-rio_amd")


@pytest.fixture(scope="session")
def oracle_apd():
    import oracle
    from oracle import apd

    oracle.build()
    return apd


@pytest.fixture(scope="session")
def gpu(gorio):
    """A usable HIP device through the product library.  A missing library is an ERROR (never a skip, never a fallback);
    only the genuine absence of a GPU (this authoring container) skips."""
    lib = gorio.load_library()  # raises GorioError when libgorio_amd.so has not been built
    import ctypes as C

    h = C.c_void_p()
    rc = lib.gorio_apd_create(C.byref(h), 0)
    if rc == -2:
        pytest.skip("no HIP device visible (GORIO_ERR_NO_DEVICE)")
    assert rc == 0, f"gorio_apd_create failed with {rc}"
    lib.gorio_apd_destroy(h)
    return 0


def rot_err(Ta, Tb):
    """(translation error [m], rotation error [rad]) between two 4x4 transforms."""
    d = np.linalg.inv(np.asarray(Ta, float)) @ np.asarray(Tb, float)
    R = d[:3, :3]
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    ang = np.arctan2(np.linalg.norm(v), (np.trace(R) - 1.0) / 2.0)  # well conditioned near 0 (arccos of the trace is not)
    return float(np.linalg.norm(d[:3, 3])), float(ang)


@pytest.fixture(scope="session")
def pose_err():
    return rot_err
