import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """The tests need the product library and the oracle; build them when a fresh checkout has neither
    (`__graft_entry__.build()` is incremental: a no-op when the binaries are current).  On the GPU box the
    prebuilt binaries travel with the snapshot; a missing hipcc there is an error, not a skip."""
    import __graft_entry__ as entry

    if not os.path.exists(entry.LIB) or not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libtpsoracle.so")):
        entry.build()
    yield
