import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # built artefacts are kept out of git: in a fresh checkout build them once (hipcc cross-compiles without a GPU)
    import glob
    have_lib = os.path.exists(os.path.join(ROOT, "cp2_amd", "lib", "libcp2hip.so"))
    have_ext = bool(glob.glob(os.path.join(ROOT, "cp2_amd", "_autograd_ext*.so")))
    if not (have_lib and have_ext):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
