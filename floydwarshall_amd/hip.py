"""Minimal ctypes binding of the HIP runtime: device buffers, streams and events WITHOUT torch.

Why it exists.  libfwx's C ABI takes plain device pointers and a hipStream_t; a host that is not a
torch program (the Haskell shim, the C consumer, the N=1 benchmark) should not have to load
PyTorch just to own two buffers and a stream.  It also keeps ONE HIP runtime in the process: torch
wheels bundle a private libamdhip64 next to /opt/rocm's, and a profiler (rocprofv3) preloads the
latter -- with torch imported, both are mapped.  A process that never imports torch has exactly the
runtime libfwx itself is linked against.

Only what the benchmark and the tests need is bound.  Everything raises HipError on failure.
"""
import ctypes

import numpy as np

from . import _lib

_HIP = None

hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault = 1, 2, 3, 4
hipStreamNonBlocking = 1


class HipError(RuntimeError):
    pass


def rt():
    """The libamdhip64 instance libfwx is bound to (loaded by SONAME, so it is the same mapping)."""
    global _HIP
    if _HIP is None:
        _lib.lib()                       # maps libfwx and, through its DT_NEEDED, the HIP runtime
        h = ctypes.CDLL("libamdhip64.so.7")
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        sigs = {
            "hipGetDeviceCount": [ctypes.POINTER(ci)], "hipSetDevice": [ci],
            "hipGetDevice": [ctypes.POINTER(ci)], "hipDeviceSynchronize": [],
            "hipMalloc": [ctypes.POINTER(vp), sz], "hipFree": [vp],
            "hipMemcpy": [vp, vp, sz, ci], "hipMemcpyAsync": [vp, vp, sz, ci, vp],
            "hipMemsetAsync": [vp, ci, sz, vp], "hipMemsetD32Async": [vp, ci, sz, vp],
            "hipStreamCreateWithFlags": [ctypes.POINTER(vp), ctypes.c_uint], "hipStreamDestroy": [vp],
            "hipStreamSynchronize": [vp],
            "hipEventCreate": [ctypes.POINTER(vp)], "hipEventDestroy": [vp],
            "hipEventRecord": [vp, vp], "hipEventSynchronize": [vp],
            "hipEventElapsedTime": [ctypes.POINTER(ctypes.c_float), vp, vp],
            "hipMemGetInfo": [ctypes.POINTER(sz), ctypes.POINTER(sz)],
        }
        for name, args in sigs.items():
            fn = getattr(h, name)
            fn.restype = ci
            fn.argtypes = args
        h.hipGetErrorString.restype = ctypes.c_char_p
        h.hipGetErrorString.argtypes = [ci]
        _HIP = h
    return _HIP


def _ck(err, what):
    if err != 0:
        raise HipError("%s: %s (hipError_t %d)" % (what, rt().hipGetErrorString(err).decode(), err))


def device_count():
    c = ctypes.c_int(0)
    return c.value if rt().hipGetDeviceCount(ctypes.byref(c)) == 0 else 0


def set_device(i):
    _ck(rt().hipSetDevice(int(i)), "hipSetDevice")


def synchronize(devices=None):
    """hipDeviceSynchronize on the current device, or on each device of `devices` (the current device
    is restored)."""
    if devices is None:
        _ck(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")
        return
    cur = ctypes.c_int(0)
    _ck(rt().hipGetDevice(ctypes.byref(cur)), "hipGetDevice")
    for d in sorted(set(devices)):
        set_device(d)
        _ck(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")
    set_device(cur.value)


def mem_get_info():
    """(free, total) bytes of the current device."""
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _ck(rt().hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)), "hipMemGetInfo")
    return int(free.value), int(total.value)


_DEFAULT_STREAM = None


def default_stream():
    """The stream engine.py's device-pointer API launches on when the caller names none: one
    process-wide non-blocking stream (never the legacy null stream)."""
    global _DEFAULT_STREAM
    if _DEFAULT_STREAM is None:
        _DEFAULT_STREAM = Stream()
    return _DEFAULT_STREAM


class Stream:
    """A non-blocking stream (never the legacy null stream)."""

    def __init__(self):
        h = ctypes.c_void_p()
        _ck(rt().hipStreamCreateWithFlags(ctypes.byref(h), hipStreamNonBlocking), "hipStreamCreate")
        self.ptr = h

    def synchronize(self):
        _ck(rt().hipStreamSynchronize(self.ptr), "hipStreamSynchronize")

    def close(self):
        if self.ptr:
            rt().hipStreamDestroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Event:
    def __init__(self):
        h = ctypes.c_void_p()
        _ck(rt().hipEventCreate(ctypes.byref(h)), "hipEventCreate")
        self.ptr = h

    def record(self, stream):
        _ck(rt().hipEventRecord(self.ptr, stream.ptr if stream is not None else None), "hipEventRecord")

    def synchronize(self):
        _ck(rt().hipEventSynchronize(self.ptr), "hipEventSynchronize")

    def elapsed_time(self, later):
        """Milliseconds from this event to `later` (both must have completed)."""
        ms = ctypes.c_float(0.0)
        _ck(rt().hipEventElapsedTime(ctypes.byref(ms), self.ptr, later.ptr), "hipEventElapsedTime")
        return float(ms.value)

    def __del__(self):
        try:
            if self.ptr:
                rt().hipEventDestroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class DeviceArray:
    """A C-contiguous array in HBM.  Quacks enough like a torch tensor for engine.py's device API
    (data_ptr / shape / dtype / element_size / is_cuda / is_contiguous / dim); row slices are views."""

    is_cuda = True

    def __init__(self, shape, dtype, _ptr=None, _base=None):
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._base = _base
        if _ptr is None:
            p = ctypes.c_void_p()
            _ck(rt().hipMalloc(ctypes.byref(p), max(self.nbytes, 1)), "hipMalloc(%d)" % self.nbytes)
            self._ptr = p.value
            self._owned = True
        else:
            self._ptr = int(_ptr)
            self._owned = False

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.copy_from_host(a, stream)
        return d

    def data_ptr(self):
        return self._ptr

    def element_size(self):
        return self.dtype.itemsize

    def dim(self):
        return len(self.shape)

    def is_contiguous(self):
        return True

    def numel(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def rows(self, lo, hi):
        """View of rows [lo, hi) of a 2-D array."""
        assert len(self.shape) == 2 and 0 <= lo <= hi <= self.shape[0]
        pitch = self.shape[1] * self.dtype.itemsize
        return DeviceArray((hi - lo, self.shape[1]), self.dtype, _ptr=self._ptr + lo * pitch, _base=self)

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.step in (None, 1)
        lo, hi, _ = sl.indices(self.shape[0])
        return self.rows(lo, hi)

    def copy_from_host(self, a, stream=None):
        a = np.ascontiguousarray(a)
        assert a.nbytes == self.nbytes
        s = stream.ptr if stream is not None else None
        if s is None:
            _ck(rt().hipMemcpy(self._ptr, a.ctypes.data, self.nbytes, hipMemcpyHostToDevice), "hipMemcpy H2D")
        else:
            _ck(rt().hipMemcpyAsync(self._ptr, a.ctypes.data, self.nbytes, hipMemcpyHostToDevice, s),
                "hipMemcpyAsync H2D")

    def copy_(self, other, stream=None):
        """Device-to-device copy, asynchronous on `stream` (or blocking without one)."""
        assert other.nbytes == self.nbytes
        if stream is None:
            _ck(rt().hipMemcpy(self._ptr, other._ptr, self.nbytes, hipMemcpyDeviceToDevice), "hipMemcpy D2D")
        else:
            _ck(rt().hipMemcpyAsync(self._ptr, other._ptr, self.nbytes, hipMemcpyDeviceToDevice,
                                    stream.ptr), "hipMemcpyAsync D2D")
        return self

    def clone(self, stream=None):
        """A new array with the same contents (device-to-device copy on `stream` / the default stream)."""
        return DeviceArray(self.shape, self.dtype).copy_(self, stream if stream is not None
                                                         else default_stream())

    def zero_(self, stream=None):
        _ck(rt().hipMemsetAsync(self._ptr, 0, self.nbytes, stream.ptr if stream is not None else None),
            "hipMemsetAsync")
        if stream is None:
            synchronize()
        return self

    def fill_(self, value, stream=None):
        """Every element = value (32-bit element types only: int32 / float32)."""
        assert self.dtype.itemsize == 4
        word = int(np.array([value], dtype=self.dtype).view(np.int32)[0])
        st = stream if stream is not None else default_stream()
        _ck(rt().hipMemsetD32Async(self._ptr, word, self.nbytes // 4, st.ptr), "hipMemsetD32Async")
        st.synchronize()
        return self

    def numpy(self, stream=None):
        # what the default stream (or `stream`) has queued on this array is finished first; hipMemcpy
        # itself does not wait for non-blocking streams
        (stream if stream is not None else default_stream()).synchronize()
        out = np.empty(self.shape, dtype=self.dtype)
        _ck(rt().hipMemcpy(out.ctypes.data, self._ptr, self.nbytes, hipMemcpyDeviceToHost), "hipMemcpy D2H")
        return out

    def free(self):
        if self._owned and self._ptr:
            rt().hipFree(self._ptr)
            self._ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
