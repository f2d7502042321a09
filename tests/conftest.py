import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hiplib():
    """The product C-ABI library.  Built in-tree if missing (hipcc cross-compiles without a GPU)."""
    import librempeg_amd
    if not os.path.exists(librempeg_amd.library_path()):
        librempeg_amd.build_library()
    return librempeg_amd.load_library()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
