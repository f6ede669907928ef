import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# dmabuf IPC (RCCL between devices and device memory shared across processes need it on this pool): set
# before the first HIP call of the pytest process and inherited by every rank the tests spawn.  The
# loader (floydwarshall_amd/_lib.py) sets the same default; here it also covers spawned helpers that
# import torch first.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


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


@pytest.fixture(autouse=True)
def _disarm_the_fault_injection_hook():
    """fwx_test_fail_after arms a thread-local countdown; a test that fails between arming it and its own
    clean-up must not make later, unrelated tests see FWX_ERR_OOM."""
    yield
    from floydwarshall_amd import _lib
    _lib.lib().fwx_test_fail_after(0)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


