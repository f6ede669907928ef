"""The C-ABI library loads and exports every symbol include/*.h declares.  CPU only: no compute
call is made (there is no GPU here); the entry points must report FWX_ERR_NO_DEVICE, never fall
back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

from floydwarshall_amd import _lib, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b(fwxh?_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(_lib.LIB_PATH)
    headers = [h for h in os.listdir(os.path.join(ROOT, "include")) if h.endswith(".h")]
    assert "fwx.h" in headers
    names = []
    for h in headers:
        names += _declared_functions(h)
    assert len(names) >= 16
    for name in names:
        assert hasattr(L, name), "include/ declares %s but libfwx.so does not export it" % name


def test_binding_table_covers_the_engine_header():
    assert sorted(_lib.SIGNATURES) == _declared_functions("fwx.h")


def test_abi_version_and_strerror():
    L = _lib.lib()
    assert L.fwx_abi_version() == 1
    assert L.fwx_strerror(0) == b"ok"
    assert b"no HIP device" in L.fwx_strerror(_lib.FWX_ERR_NO_DEVICE)
    assert L.fwx_device_count() >= 0


def test_no_cuda_symbols_or_torch_in_the_abi():
    # plain pointers and sizes only: the library must not depend on torch or python
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "python" not in out
    assert "amdhip64" in out


def test_follow_path_host_side():
    nxt = np.array([[-1, 1, 1], [2, -1, 2], [0, 0, -1]], dtype=np.int32)
    assert engine.follow_path(nxt, 0, 2) == [1, 2]
    assert engine.follow_path(nxt, 0, 0) == []
    loop = np.array([[-1, 1, 1], [0, -1, 0], [0, 0, -1]], dtype=np.int32)  # 0->1->0->... never 2
    with pytest.raises(engine.FwxError) as e:
        engine.follow_path(loop, 0, 2)
    assert e.value.status == _lib.FWX_ERR_CYCLE


@pytest.mark.skipif(engine.device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_device():
    rate = np.ones((4, 4))
    before = rate.copy()
    with pytest.raises(engine.FwxError) as e:
        engine.solve(rate)
    assert e.value.status == _lib.FWX_ERR_NO_DEVICE
    assert np.array_equal(rate, before)          # untouched: nothing was computed anywhere
    with pytest.raises(engine.FwxError):
        engine.DeviceMatrix(4)
    engine.solve(np.zeros((0, 0)))                # n == 0 is success and touches nothing


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "floydwarshall_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "fworacle" not in text and "fwo_" not in text, f
