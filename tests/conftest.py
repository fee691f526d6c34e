import os
import sys

import numpy as np
import pytest

try:
    # torch bundles its own ROCm runtime; libngravs_hip.so links /opt/rocm's.  One process must end up with ONE HIP runtime:
    # whichever library is loaded first decides which, and torch only initialises with its own -- so torch goes first
    # (bench.py does the same; C hosts never load torch).
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def O():
    o = ge.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def have_lib(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        ge.build()
    return pkg.lib()


def galaxy_ic(pkg):
    p = os.path.join(ROOT, "tests", "golden", "GalaxyCollision.IC")
    return pkg.ic.read_gadget_format1(p)


def galaxy_config(pkg, **kw):
    return pkg.make_config(n_gravs=2, G=43007.1, theta=0.5, softening=[0, 1.0, 0.4, 1.0, 1.0, 1.0],
                           type_to_grav=[0, 0, 1, 0, 0, 0], wiring="newton", tree_alloc_factor=0.8, **kw)


def rel_err(a, b):
    """|a-b| / |b| per row"""
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
