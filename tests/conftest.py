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


@pytest.fixture(scope='session')
def built_lib():
    """ make sure libbild_amd.so exists (cross-compiles without a GPU) """
    import __graft_entry__ as entry
    entry.build()
    from bild_amd import _lib
    return _lib.lib()
