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
    assert L.fwx_abi_version() == 3 == _lib.FWX_ABI_VERSION
    assert b"RCCL" in L.fwx_strerror(_lib.FWX_ERR_RCCL) and b"exception" in L.fwx_strerror(_lib.FWX_ERR_INTERNAL)
    assert L.fwx_strerror(0) == b"ok"
    assert b"no HIP device" in L.fwx_strerror(_lib.FWX_ERR_NO_DEVICE)
    assert L.fwx_device_count() >= 0


def test_no_cuda_symbols_or_torch_in_the_abi():
    # plain pointers and sizes only: the library must not depend on torch or python
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "python" not in out
    assert "amdhip64" in out


def test_libfwx_runs_on_the_hip_runtime_it_was_built_against():
    """The loader imports nothing besides libfwx (no torch), so the library's DT_NEEDED resolves
    through its RUNPATH to /opt/rocm's libamdhip64 -- the runtime it was compiled against.  (A process
    that imports torch FIRST gets torch's bundled runtime instead and a RuntimeWarning saying so.)"""
    import sys
    assert "torch" not in sys.modules, "the pytest process must stay torch-free (helpers.spawn_ranks)"
    built, runtime, same = _lib.runtime_versions()
    assert built > 0
    assert runtime == 0 or same, (built, runtime)       # 0: no HIP runtime initialisable (CPU box)
    maps = open("/proc/self/maps").read()
    hip_libs = {ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}
    assert len(hip_libs) == 1 and "torch" not in next(iter(hip_libs)), hip_libs


def test_exception_barrier_without_a_device():
    """fwx_test_fail_after arms a bad_alloc at the next internal allocation point of this thread;
    the boundary must turn it into FWX_ERR_OOM (host mirror entry point: needs no GPU)."""
    from floydwarshall_amd import host
    L = _lib.lib()
    s = host.Session()
    try:
        devs = (ctypes.c_int32 * 2)(0, 0)
        assert L.fwx_test_fail_after(1) == _lib.FWX_OK
        assert host.hlib().fwxh_session_set_devices(s._h, 2, devs, 0) == _lib.FWX_ERR_OOM
        assert host.hlib().fwxh_session_set_devices(s._h, 2, devs, 0) == _lib.FWX_OK   # disarmed again
        assert L.fwx_test_fail_after(0) == _lib.FWX_OK
    finally:
        s.close()


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


def test_argument_validation_without_touching_a_device():
    """Bad arguments are rejected with FWX_ERR_INVALID before any device work (the reference's hot
    path is total: it has no failure modes of its own to mirror, so the ABI defines them)."""
    L = _lib.lib()
    one = np.ones((2, 2))
    assert L.fwx_solve_f64(-1, one.ctypes.data, None, None, None) == _lib.FWX_ERR_INVALID
    assert L.fwx_solve_f64(0, None, None, None, None) == _lib.FWX_OK            # empty matrix
    hops = np.ones((2, 2), dtype=np.int32)
    assert L.fwx_solve_f64(2, one.ctypes.data, None, hops.ctypes.data, None) == _lib.FWX_ERR_INVALID
    o = _lib.FwxOpts()
    o.struct_size = 4                                                            # too small
    assert L.fwx_solve_f64(2, one.ctypes.data, None, None, ctypes.byref(o)) == _lib.FWX_ERR_INVALID
    o.struct_size = ctypes.sizeof(_lib.FwxOpts)
    o.k_begin, o.k_end = 3, 2
    assert L.fwx_solve_f64(2, one.ctypes.data, None, None, ctypes.byref(o)) == _lib.FWX_ERR_INVALID
    o.k_begin, o.k_end, o.engine = 0, 0, 99
    assert L.fwx_solve_f64(2, one.ctypes.data, None, None, ctypes.byref(o)) == _lib.FWX_ERR_INVALID
    s = _lib.FwxSlab()
    s.n, s.row0, s.rows, s.dtype = 4, 3, 2, _lib.FWX_F32                         # rows past the end
    p = _lib.FwxPivots()
    assert L.fwx_dev_relax(ctypes.byref(s), ctypes.byref(p), 1, None, None) == _lib.FWX_ERR_INVALID
    assert L.fwx_dev_relax(None, None, 1, None, None) == _lib.FWX_ERR_INVALID
    nxt = np.full((2, 2), -1, dtype=np.int32)
    assert L.fwx_follow_path(2, nxt.ctypes.data, 0, 5, None, 0) == _lib.FWX_ERR_INVALID
    assert L.fwx_follow_path(2, nxt.ctypes.data, 0, 1, None, 0) == 0            # no route
    assert L.fwx_matrix_create(None, 4, 0, 1, 0, -1) == _lib.FWX_ERR_INVALID
    assert L.fwx_matrix_destroy(None) == _lib.FWX_OK
    assert b"unknown" in L.fwx_strerror(-999)
