import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def entry():
    import __graft_entry__ as G
    lib_path = os.path.join(G.PKG_DIR, "lib", "librusty_marcher_amd.so")
    if not os.path.exists(lib_path) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        G.build()
    return G


@pytest.fixture(scope="session")
def pkg(entry):
    return entry.load_package()


@pytest.fixture(scope="session")
def O(entry):
    return entry.load_oracle()


@pytest.fixture(scope="session")
def golden_ppm():
    """The reference's committed engine/out.ppm (800x600 P6), stored gzipped."""
    with gzip.open(os.path.join(GOLDEN, "out_800x600.ppm.gz"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def cornell_path():
    return os.path.join(GOLDEN, "cornell_box.obj")
