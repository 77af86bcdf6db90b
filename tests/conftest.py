import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope='session', autouse=True)
def _build_once():
    """
    libbild_amd.so (cross-compiles without a GPU) and the oracle exist before any test runs: the samplers use the
    library's host-side bookkeeping even in tests that never touch a GPU.  A no-op when everything is up to date.
    """
    import __graft_entry__ as entry
    entry.build()


@pytest.fixture(scope='session')
def built_lib(_build_once):
    """ the loaded library handle """
    from bild_amd import _lib
    return _lib.lib()
