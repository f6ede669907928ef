import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build (or refresh) the native pieces before collection: libfwx.so + fwx_cli with hipcc
    # (cross-compiles without a GPU) and the oracle with gcc.  The .so files are git-ignored, so a
    # fresh checkout has none, and some test modules query the library while being imported.
    from floydwarshall_amd import build as fbuild
    fbuild.build_lib()
    import oracle
    oracle.build()
    oracle.lib()
    # Map libfwx NOW, before any test module is imported: the pytest process never imports torch
    # (ranks that need it are spawned, helpers.spawn_ranks), so libfwx is bound to the HIP runtime it
    # was built against (/opt/rocm) -- tests/test_abi_symbols.py checks exactly that.
    from floydwarshall_amd import _lib
    _lib.lib()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


