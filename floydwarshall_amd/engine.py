"""Python binding over the libfwx C ABI (include/fwx.h).

This is plumbing for the tests, the benchmark and the multi-GPU driver: numpy arrays for the
host-buffer entry points, raw device pointers for the device step API.  Device memory is owned by
floydwarshall_amd.hip.DeviceArray (the HIP runtime through ctypes: no torch in the process, so libfwx
runs on the runtime it was built against) or, for floydwarshall_amd.dist, by torch tensors; every
function here takes either and allocates its outputs like its inputs.
The functions mirror the reference's seam, floydWarshall = runAlgo 0 . buildMatrix
(/root/reference/src/lib/Algorithms.hs:19-20): `solve` IS runAlgo on the dense form.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (FWX_ENGINE_AUTO, FWX_ENGINE_FUSED, FWX_ENGINE_PERK, FWX_F32, FWX_F64,
                   FWX_FUSED_BLOCK, FWX_UPDATE_SHARDS, FWX_XCHG_AUTO, FWX_XCHG_PEER, FWX_XCHG_RCCL,
                   FWX_ERR_RCCL, FwxError, FwxOpts, FwxPivots, FwxSlab, check, lib)

__all__ = ["solve", "follow_path", "dev_follow_paths", "dev_check_nonneg", "dev_domain_bits", "dev_solve_fused",
           "dev_solve", "DeviceMatrix", "dev_relax", "dev_panel", "dev_panel_snap",
           "dev_relax_fused", "FusedWorkspace", "Trace", "FWX_FUSED_BLOCK", "device_count",
           "FwxError", "FWX_ENGINE_AUTO", "FWX_ENGINE_PERK", "FWX_ENGINE_FUSED",
           "FWX_UPDATE_SHARDS", "solve_multi", "FWX_XCHG_AUTO", "FWX_XCHG_PEER", "FWX_XCHG_RCCL", "FWX_ERR_RCCL"]


def device_count():
    return lib().fwx_device_count()


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _check_arrays(rate, nxt, hops):
    if not (isinstance(rate, np.ndarray) and rate.ndim == 2 and rate.shape[0] == rate.shape[1]):
        raise ValueError("rate must be a square 2-D numpy array")
    if rate.dtype not in (np.float32, np.float64) or not rate.flags.c_contiguous:
        raise ValueError("rate must be C-contiguous float32 or float64")
    for name, a in (("next", nxt), ("hops", hops)):
        if a is not None and not (isinstance(a, np.ndarray) and a.shape == rate.shape
                                  and a.dtype == np.int32 and a.flags.c_contiguous):
            raise ValueError("%s must be a C-contiguous int32 array shaped like rate" % name)
    if hops is not None and nxt is None:
        raise ValueError("hops requires next")


def _opts(device=-1, engine=FWX_ENGINE_AUTO, k_begin=0, k_end=0, block=0, serpentine=True,
          want_updates=False, stream=None):
    """fwx_opts.  stream: a hip.Stream / torch stream handle the (still blocking) call runs on instead
    of a library-owned non-blocking stream."""
    o = FwxOpts()
    o.struct_size = ctypes.sizeof(FwxOpts)
    o.device, o.engine, o.k_begin, o.k_end, o.block = device, engine, k_begin, k_end, block
    o.serpentine = 0 if serpentine else 1
    u = ctypes.c_uint64(0)
    if want_updates:
        o.updates_out = ctypes.pointer(u)
    if stream is not None:
        o.stream = _stream_ptr(stream)
        o.use_stream = 1
    return o, u


def solve(rate, nxt=None, hops=None, *, device=-1, engine=FWX_ENGINE_AUTO, k_begin=0, k_end=0,
          block=0, serpentine=True, count_updates=False):
    """runAlgo (Algorithms.hs:42-61) in place on host numpy arrays, on the GPU.

    Returns U (number of successful relaxations) if count_updates else None."""
    _check_arrays(rate, nxt, hops)
    o, u = _opts(device, engine, k_begin, k_end, block, serpentine, count_updates)
    fn = lib().fwx_solve_f64 if rate.dtype == np.float64 else lib().fwx_solve_f32
    check(fn(rate.shape[0], _np_ptr(rate), _np_ptr(nxt), _np_ptr(hops), ctypes.byref(o)),
          "fwx_solve")
    return int(u.value) if count_updates else None


def solve_multi(rate, nxt=None, hops=None, *, devices=(0,), exchange=FWX_XCHG_AUTO,
                engine=FWX_ENGINE_AUTO, k_begin=0, k_end=0, count_updates=False):
    """runAlgo in place on host numpy arrays, row-partitioned over `devices` (one partition per
    entry; a device may repeat = logical partitions on one GPU) from ONE process:
    fwx_solve_multi_f64 / _f32."""
    _check_arrays(rate, nxt, hops)
    o, u = _opts(engine=engine, k_begin=k_begin, k_end=k_end, want_updates=count_updates)
    devs = (ctypes.c_int32 * len(devices))(*devices)
    fn = lib().fwx_solve_multi_f64 if rate.dtype == np.float64 else lib().fwx_solve_multi_f32
    check(fn(rate.shape[0], _np_ptr(rate), _np_ptr(nxt), _np_ptr(hops), len(devices), devs, exchange,
             ctypes.byref(o)), "fwx_solve_multi")
    return int(u.value) if count_updates else None


def follow_path(nxt, src, dst):
    """Vertex indices after src up to and including dst; [] when there is no route."""
    n = nxt.shape[0]
    out = np.empty(max(n, 1), dtype=np.int32)
    ln = check(lib().fwx_follow_path(n, _np_ptr(nxt), int(src), int(dst), _np_ptr(out), n),
               "fwx_follow_path")
    return [int(x) for x in out[:ln]]


class DeviceMatrix:
    """fwx_matrix handle: the solved matrix stays in HBM across queries (InSync, Types.hs:35-37)."""

    def __init__(self, n, dtype=np.float64, with_next=True, with_hops=False, device=-1, devices=None,
                 exchange=FWX_XCHG_AUTO):
        """devices: a list of HIP ordinals = a ROW-PARTITIONED handle (fwx_matrix_create_multi), one
        partition per entry, repeats allowed; None = one device."""
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.with_next, self.with_hops = bool(with_next), bool(with_hops)
        h = ctypes.c_void_p()
        code = FWX_F64 if self.dtype == np.float64 else FWX_F32
        if devices is None:
            check(lib().fwx_matrix_create(ctypes.byref(h), self.n, code, int(self.with_next),
                                          int(self.with_hops), device), "fwx_matrix_create")
        else:
            devs = (ctypes.c_int32 * len(devices))(*devices)
            check(lib().fwx_matrix_create_multi(ctypes.byref(h), self.n, code, int(self.with_next),
                                                int(self.with_hops), len(devices), devs, exchange),
                  "fwx_matrix_create_multi")
        self._h = h

    def parts(self):
        """(number of row partitions, exchange transport in use)."""
        x = ctypes.c_int32(0)
        p = check(lib().fwx_matrix_parts(self._h, ctypes.byref(x)), "fwx_matrix_parts")
        return p, int(x.value)

    def part_rows(self, part):
        """(row0, rows) of partition `part` (fwx_matrix_part_rows)."""
        r0, rs = ctypes.c_int32(0), ctypes.c_int32(0)
        check(lib().fwx_matrix_part_rows(self._h, int(part), ctypes.byref(r0), ctypes.byref(rs)), "fwx_matrix_part_rows")
        return int(r0.value), int(rs.value)

    def comm_ranks(self):
        """Ranks of the RCCL communicator the partitions exchange panels on (0: not RCCL)."""
        return check(lib().fwx_matrix_comm_ranks(self._h), "fwx_matrix_comm_ranks")

    def set_timing(self, on=True):
        """Per-step event timings of the following solves (partitioned handles; fwx_matrix_set_timing)."""
        check(lib().fwx_matrix_set_timing(self._h, 1 if on else 0), "fwx_matrix_set_timing")

    def timing(self):
        """fwx_multi_timing of the last solve as a dict (microseconds; see include/fwx.h)."""
        from ._lib import FwxMultiTiming
        t = FwxMultiTiming()
        t.struct_size = ctypes.sizeof(FwxMultiTiming)
        check(lib().fwx_matrix_get_timing(self._h, ctypes.byref(t)), "fwx_matrix_get_timing")
        return {name: getattr(t, name) for name, _ in FwxMultiTiming._fields_ if name != "struct_size"}

    def upload(self, rate, nxt=None, hops=None):
        _check_arrays(rate, nxt, hops)
        assert rate.shape[0] == self.n and rate.dtype == self.dtype
        check(lib().fwx_matrix_upload(self._h, _np_ptr(rate), _np_ptr(nxt), _np_ptr(hops)),
              "fwx_matrix_upload")

    def upload_dev(self, rate_d, next_d=None, hops_d=None):
        """Upload from DEVICE arrays (torch tensors or hip.DeviceArray): fwx_matrix_upload takes host
        or device pointers.  The caller must have finished writing them (this call blocks)."""
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        check(lib().fwx_matrix_upload(self._h, ptr(rate_d), ptr(next_d), ptr(hops_d)),
              "fwx_matrix_upload")

    def download_dev(self, rate_d=None, next_d=None, hops_d=None):
        """Copy the handle's arrays into DEVICE arrays."""
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        check(lib().fwx_matrix_download(self._h, ptr(rate_d), ptr(next_d), ptr(hops_d)),
              "fwx_matrix_download")

    def solve(self, **kw):
        count = kw.pop("count_updates", False)
        o, u = _opts(want_updates=count, **kw)
        check(lib().fwx_matrix_solve(self._h, ctypes.byref(o)), "fwx_matrix_solve")
        return int(u.value) if count else None

    def download(self):
        rate = np.empty((self.n, self.n), dtype=self.dtype)
        nxt = np.empty((self.n, self.n), dtype=np.int32) if self.with_next else None
        hops = np.empty((self.n, self.n), dtype=np.int32) if self.with_hops else None
        check(lib().fwx_matrix_download(self._h, _np_ptr(rate), _np_ptr(nxt), _np_ptr(hops)),
              "fwx_matrix_download")
        return rate, nxt, hops

    def query(self, src, dst):
        """(rate, [path indices]) of one entry, read back from the device."""
        r = ctypes.c_double(0.0)
        out = np.empty(max(self.n, 1), dtype=np.int32)
        ln = check(lib().fwx_matrix_query(self._h, int(src), int(dst), ctypes.byref(r),
                                          _np_ptr(out), self.n), "fwx_matrix_query")
        return float(r.value), [int(x) for x in out[:ln]]

    def keep_input(self):
        """Keep the uploaded input on the device (fwx_matrix_keep_input) so that patch_input can
        replace a few entries of it without a new upload."""
        check(lib().fwx_matrix_keep_input(self._h), "fwx_matrix_keep_input")

    def patch_input(self, index, rate_vals, next_vals=None, hops_vals=None):
        """Replace entries index[q] = i*n + j of the KEPT input and make it the unsolved matrix again
        (fwx_matrix_patch_input); follow with solve()."""
        index = np.ascontiguousarray(index, dtype=np.int64)
        rate_vals = np.ascontiguousarray(rate_vals, dtype=self.dtype)
        nv = None if next_vals is None else np.ascontiguousarray(next_vals, dtype=np.int32)
        hv = None if hops_vals is None else np.ascontiguousarray(hops_vals, dtype=np.int32)
        assert index.ndim == 1 and rate_vals.shape == index.shape
        check(lib().fwx_matrix_patch_input(self._h, len(index), _np_ptr(index), _np_ptr(rate_vals),
                                           _np_ptr(nv), _np_ptr(hv)), "fwx_matrix_patch_input")

    def enable_resume(self, checkpoints=7):
        """Keep state checkpoints + the panels of every pivot so that resolve() can resume
        (fwx_matrix_enable_resume); returns the number of checkpoints placed."""
        return check(lib().fwx_matrix_enable_resume(self._h, int(checkpoints)), "fwx_matrix_enable_resume")

    def resolve(self, index, rate_vals, next_vals=None, hops_vals=None, **kw):
        """patch_input + solve in one call, resuming from the last checkpoint the changed entries
        cannot have influenced (fwx_matrix_resolve).  Returns the pivot the solve started at."""
        index = np.ascontiguousarray(index, dtype=np.int64)
        rate_vals = np.ascontiguousarray(rate_vals, dtype=self.dtype)
        nv = None if next_vals is None else np.ascontiguousarray(next_vals, dtype=np.int32)
        hv = None if hops_vals is None else np.ascontiguousarray(hops_vals, dtype=np.int32)
        assert index.ndim == 1 and rate_vals.shape == index.shape
        o, _u = _opts(want_updates=kw.pop("count_updates", False), **kw)   # _u: keeps updates_out alive
        started = ctypes.c_int32(0)
        check(lib().fwx_matrix_resolve(self._h, len(index), _np_ptr(index), _np_ptr(rate_vals), _np_ptr(nv),
                                       _np_ptr(hv), ctypes.byref(o), ctypes.byref(started)),
              "fwx_matrix_resolve")
        return int(started.value)

    def enable_path_log(self):
        """Keep the path trace (three n x n int32 matrices, see fwx.h) so that query_exact can
        rebuild the reference's `_path` lists exactly; a traced solve() needs a fresh upload()."""
        check(lib().fwx_matrix_enable_path_log(self._h), "fwx_matrix_enable_path_log")

    def path_log_count(self):
        c = ctypes.c_uint64(0)
        check(lib().fwx_matrix_path_log_count(self._h, ctypes.byref(c)), "fwx_matrix_path_log_count")
        return int(c.value)

    def query_exact(self, src, dst, cap=None):
        """(rate, the reference's `_path` as vertex indices) -- exact under ties."""
        cap = cap or max(4 * self.n, 64)
        r = ctypes.c_double(0.0)
        out = np.empty(cap, dtype=np.int32)
        ln = check(lib().fwx_matrix_query_exact(self._h, int(src), int(dst), ctypes.byref(r),
                                                _np_ptr(out), cap), "fwx_matrix_query_exact")
        return float(r.value), [int(x) for x in out[:ln]]

    def query_exact_batch(self, src, dst, cap=None):
        """Exact `_path` lists of many (src[q], dst[q]) pairs in one launch -> list of lists
        (FwxError if one does not fit into cap entries)."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        assert src.shape == dst.shape and src.ndim == 1
        cap = cap or max(4 * self.n, 64)
        lens = np.empty(len(src), dtype=np.int32)
        paths = np.empty((len(src), cap), dtype=np.int32)
        check(lib().fwx_matrix_query_exact_batch(self._h, len(src), _np_ptr(src), _np_ptr(dst),
                                                 _np_ptr(lens), _np_ptr(paths), cap),
              "fwx_matrix_query_exact_batch")
        if len(lens) and int(lens.min()) < 0:
            raise FwxError(int(lens.min()), "fwx_matrix_query_exact_batch: a list did not fit")
        return [[int(v) for v in paths[q, :lens[q]]] for q in range(len(src))]

    def close(self):
        if self._h:
            lib().fwx_matrix_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- device-pointer step API (torch tensors own the memory) -----------------------------------

def _dtype_name(t):
    """'float32' for torch.float32 and numpy float32 alike: the device API takes torch tensors or
    hip.DeviceArray (no torch in the process)."""
    return str(t.dtype).split(".")[-1]


def _tensor_dtype_code(t):
    name = _dtype_name(t)
    if name == "float32":
        return FWX_F32
    if name == "float64":
        return FWX_F64
    raise ValueError("rate tensor must be float32 or float64")


def _slab(rate_t, next_t, hops_t, n, row0):
    assert rate_t.is_cuda and rate_t.is_contiguous() and rate_t.dim() == 2 and rate_t.shape[1] == n
    for t in (next_t, hops_t):
        assert t is None or (t.is_cuda and t.is_contiguous() and _dtype_name(t) == "int32"
                             and tuple(t.shape) == tuple(rate_t.shape))
    s = FwxSlab()
    s.n, s.row0, s.rows, s.dtype = n, row0, rate_t.shape[0], _tensor_dtype_code(rate_t)
    s.rate = rate_t.data_ptr()
    s.next = next_t.data_ptr() if next_t is not None else None
    s.hops = hops_t.data_ptr() if hops_t is not None else None
    return s


def _is_torch(t):
    return type(t).__module__.split(".")[0] == "torch"


def _stream_ptr(stream=None, like=None):
    """hipStream_t for a launch: an explicit hip.Stream / torch stream (or raw handle); else the
    current torch stream if `like` is a torch tensor, else this module's default hip.Stream."""
    if stream is not None:
        h = getattr(stream, "ptr", None)
        if h is None:
            h = getattr(stream, "cuda_stream", stream)
        return h if isinstance(h, ctypes.c_void_p) else ctypes.c_void_p(h)
    if like is not None and _is_torch(like):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(like.device).cuda_stream)
    from . import hip
    return hip.default_stream().ptr


def _empty_like_device(like, shape, dtype_name):
    """Uninitialised device array on the device of `like` (torch tensor -> torch tensor, hip.DeviceArray
    -> hip.DeviceArray)."""
    if _is_torch(like):
        import torch
        return torch.empty(tuple(shape), dtype=getattr(torch, dtype_name), device=like.device)
    from . import hip
    return hip.DeviceArray(shape, np.dtype(dtype_name))


def dev_relax(rate_t, n, row0, k_begin, k_end, *, pivots_t=None, pivot_hops_t=None, next_t=None,
              hops_t=None, serpentine=True, updates_t=None, skip=None, pivot_next_t=None,
              stream=None):
    """Apply pivots [k_begin,k_end) to the slab `rate_t` (rows [row0,row0+rows) of the n x n
    matrix) on torch's current stream, asynchronously.  skip = (lo, hi): slab rows [lo, hi)
    (multiples of 4) are left alone -- a look-ahead step has relaxed them already.

    pivots_t None: the slab holds the pivot rows itself (single-GPU solve).  Otherwise pivots_t is
    the (k_end-k_begin) x n panel of time-k snapshots from dev_panel."""
    s = _slab(rate_t, next_t, hops_t, n, row0)
    p = FwxPivots()
    p.k_begin, p.k_end = k_begin, k_end
    if pivots_t is None:
        assert row0 <= k_begin and k_end <= row0 + rate_t.shape[0]
        es = rate_t.element_size()
        p.rate = rate_t.data_ptr() + (k_begin - row0) * n * es
        p.hops = hops_t.data_ptr() + (k_begin - row0) * n * 4 if hops_t is not None else None
        p.next = next_t.data_ptr() + (k_begin - row0) * n * 4 if next_t is not None else None
        p.stride = n
    else:
        assert pivots_t.is_cuda and pivots_t.is_contiguous() and pivots_t.dtype == rate_t.dtype
        assert tuple(pivots_t.shape) == (k_end - k_begin, n)
        p.rate = pivots_t.data_ptr()
        p.hops = pivot_hops_t.data_ptr() if pivot_hops_t is not None else None
        p.next = pivot_next_t.data_ptr() if pivot_next_t is not None else None
        p.stride = n
    upd = ctypes.c_void_p(updates_t.data_ptr()) if updates_t is not None else None
    lo, hi = skip if skip else (0, 0)
    check(lib().fwx_dev_relax_skip(ctypes.byref(s), ctypes.byref(p), int(bool(serpentine)), upd,
                                   int(lo), int(hi), _stream_ptr(stream, rate_t)), "fwx_dev_relax_skip")


def dev_panel(block_rate_t, n, k0, w_rate_t, *, next_t=None, hops_t=None, w_hops_t=None,
              updates_t=None):
    """Owner-side panel phase: evolve pivot rows [k0,k0+B) in place, export time-k snapshots."""
    s = _slab(block_rate_t, next_t, hops_t, n, k0)
    assert w_rate_t.is_cuda and w_rate_t.is_contiguous() and w_rate_t.shape == block_rate_t.shape
    upd = ctypes.c_void_p(updates_t.data_ptr()) if updates_t is not None else None
    wh = ctypes.c_void_p(w_hops_t.data_ptr()) if w_hops_t is not None else None
    check(lib().fwx_dev_panel(ctypes.byref(s), ctypes.c_void_p(w_rate_t.data_ptr()), wh, upd,
                              _stream_ptr(None, block_rate_t)), "fwx_dev_panel")


def _alloc(shape, dtype, device, fill=None):
    """Device array of a torch dtype (on torch `device`) or of a numpy dtype (hip.DeviceArray on the
    current HIP device; `device` is then ignored)."""
    if type(dtype).__module__.split(".")[0] == "torch":
        import torch
        if fill is None:
            return torch.empty(tuple(shape), dtype=dtype, device=device)
        return torch.full(tuple(shape), fill, dtype=dtype, device=device)
    from . import hip
    a = hip.DeviceArray(shape, dtype)
    return a if fill is None else a.fill_(fill)


def _int32_of(dtype):
    if type(dtype).__module__.split(".")[0] == "torch":
        import torch
        return torch.int32
    return np.int32


class FusedWorkspace:
    """Device scratch of the fused engine for slabs of up to `rows` rows: snapshot panels W (two,
    for look-ahead; with hops also their hops panels WH), pivot-column snapshots Ct / CNt / CHt.
    dtype: a torch dtype (arrays are torch tensors on `device`) or a numpy dtype (hip.DeviceArray)."""

    def __init__(self, n, rows, dtype, device=None, with_next=False, with_hops=False):
        B = FWX_FUSED_BLOCK
        i32 = _int32_of(dtype)
        self.w = [_alloc((B, n), dtype, device) for _ in range(2)]
        self.wh = [_alloc((B, n), i32, device) for _ in range(2)] if with_hops else [None, None]
        ld = (max(rows, 1) + 3) & ~3
        self.ld = ld
        self.ct = _alloc((B, ld), dtype, device)
        self.cnt = _alloc((B, ld), i32, device) if with_next else None
        self.cht = _alloc((B, ld), i32, device) if with_hops else None


class Trace:
    """Path trace of a slab (fwx.h fwx_trace): three rows x n int32 device arrays, LOCAL rows, all
    -1 before a solve.  rows(lo, hi) gives the views a panel of pivot rows needs.  device: a
    torch.device (torch tensors) or None (hip.DeviceArray on the current HIP device)."""

    def __init__(self, rows, n, device=None, _views=None):
        if _views is not None:
            self.last, self.at_col, self.at_row = _views
        elif device is not None:
            import torch
            self.last, self.at_col, self.at_row = (torch.full((rows, n), -1, dtype=torch.int32, device=device)
                                                   for _ in range(3))
        else:
            self.last, self.at_col, self.at_row = (_alloc((rows, n), np.int32, None, fill=-1)
                                                   for _ in range(3))

    def rows(self, lo, hi):
        return Trace(0, 0, _views=(self.last[lo:hi], self.at_col[lo:hi], self.at_row[lo:hi]))

    def c_struct(self):
        t = _lib.FwxTrace()
        t.last, t.at_col, t.at_row = self.last.data_ptr(), self.at_col.data_ptr(), self.at_row.data_ptr()
        return t


def _trace_ref(trace):
    return ctypes.byref(trace.c_struct()) if trace is not None else None


def dev_panel_snap(block_rate_t, n, k0, w_rate_t, *, block_next_t=None, block_hops_t=None, w_hops_t=None,
                   trace=None):
    """Snapshot panel of pivot rows [k0, k0+B) (B <= 64): w[t] = row k0+t at time k0+t (and w_hops
    its hops row, if the block carries hops).  The matrix is NOT modified (contrast dev_panel).
    trace: the Trace views OF THE SAME ROWS (Trace.rows); its at_row rows are written."""
    s = _slab(block_rate_t, block_next_t, block_hops_t, n, k0)
    assert w_rate_t.is_cuda and w_rate_t.is_contiguous()
    assert tuple(w_rate_t.shape) == tuple(block_rate_t.shape) and w_rate_t.dtype == block_rate_t.dtype
    wh = None
    if block_hops_t is not None:
        assert w_hops_t is not None and tuple(w_hops_t.shape) == tuple(block_rate_t.shape)
        wh = ctypes.c_void_p(w_hops_t.data_ptr())
    check(lib().fwx_dev_panel_snap(ctypes.byref(s), ctypes.c_void_p(w_rate_t.data_ptr()), wh,
                                   _trace_ref(trace), _stream_ptr(None, block_rate_t)), "fwx_dev_panel_snap")


def dev_domain_bits(rate_t, n, row0=0, next_t=None):
    """Domain check of one slab (fwx.h "Domain"; synchronises): bit 0 = every rate is >= +0.0 and
    not NaN, bit 1 = no entry has a non-zero rate and next < 0 (always set without next_t)."""
    s = _slab(rate_t, next_t, None, n, row0)
    if _is_torch(rate_t):
        import torch
        flag = torch.full((1,), 3, dtype=torch.int32, device=rate_t.device)
        check(lib().fwx_dev_check_nonneg(ctypes.byref(s), ctypes.c_void_p(flag.data_ptr()),
                                         _stream_ptr(None, rate_t)), "fwx_dev_check_nonneg")
        return int(flag.item())
    from . import hip
    st = hip.default_stream()
    flag = hip.DeviceArray.from_numpy(np.full((1,), 3, dtype=np.int32), st)
    check(lib().fwx_dev_check_nonneg(ctypes.byref(s), ctypes.c_void_p(flag.data_ptr()), st.ptr),
          "fwx_dev_check_nonneg")
    return int(flag.numpy(st)[0])


def dev_check_nonneg(rate_t, n, row0=0):
    """True iff every rate of the slab is >= +0.0 and not NaN (synchronises)."""
    return bool(dev_domain_bits(rate_t, n, row0) & 1)


def dev_relax_fused(rate_t, n, row0, k0, k1, w_t, ws, *, next_t=None, hops_t=None, wh_t=None, trace=None,
                    updates_t=None, nonneg=False, skip=None):
    """Apply pivots [k0,k1) (at most 64) to EVERY row of the slab in one pass, from the snapshot
    panel w_t ((k1-k0) x n; wh_t: its hops panel, iff hops_t).  ws: a FusedWorkspace sized for this
    slab.  trace: the slab's Trace.  skip = (lo, hi): slab rows [lo, hi) (multiples of 8) are left
    to an earlier look-ahead step."""
    s = _slab(rate_t, next_t, hops_t, n, row0)
    p = FwxPivots()
    p.k_begin, p.k_end = k0, k1
    assert w_t.is_cuda and w_t.is_contiguous() and w_t.dtype == rate_t.dtype
    assert tuple(w_t.shape) == (k1 - k0, n)
    ld = (rate_t.shape[0] + 3) & ~3
    assert ws.ct.numel() >= FWX_FUSED_BLOCK * ld and ws.ct.dtype == rate_t.dtype
    p.rate, p.hops, p.stride = w_t.data_ptr(), None, n
    sc = _lib.FwxFusedScratch()
    sc.col_rate = ws.ct.data_ptr()
    if next_t is not None:
        assert ws.cnt is not None and ws.cnt.numel() >= FWX_FUSED_BLOCK * ld
        sc.col_next = ws.cnt.data_ptr()
    if hops_t is not None:
        assert wh_t is not None and tuple(wh_t.shape) == (k1 - k0, n) and ws.cht is not None
        p.hops = wh_t.data_ptr()
        sc.col_hops = ws.cht.data_ptr()
    upd = ctypes.c_void_p(updates_t.data_ptr()) if updates_t is not None else None
    lo, hi = skip if skip else (0, 0)
    check(lib().fwx_dev_relax_fused_skip(ctypes.byref(s), ctypes.byref(p), ctypes.byref(sc),
                                         _trace_ref(trace), upd,
                                         _lib.FWX_FLAG_NONNEG if nonneg else 0, int(lo), int(hi),
                                         _stream_ptr(None, rate_t)),
          "fwx_dev_relax_fused_skip")


def dev_solve_fused(rate_t, n, k_begin=0, k_end=None, *, next_t=None, hops_t=None, trace=None, ws=None,
                    updates_t=None, nonneg=None):
    """Single-GPU solve of pivots [k_begin,k_end) with the fused engine on a torch tensor holding
    the whole n x n matrix, one panel + one pass at a time (no look-ahead: fwx_dev_solve has it);
    asynchronous on the current stream (after one small synchronising domain check when `nonneg`
    is not given)."""
    k_end = n if k_end is None else k_end
    if nonneg is None:
        bits = dev_domain_bits(rate_t, n, 0, next_t)
        nonneg = updates_t is None and (bits == 3 if next_t is not None else bool(bits & 1))
    ws = ws or FusedWorkspace(n, n, rate_t.dtype, getattr(rate_t, "device", None),
                              with_next=next_t is not None, with_hops=hops_t is not None)
    B = FWX_FUSED_BLOCK
    for k0 in range(k_begin, k_end, B):
        k1 = min(k_end, k0 + B)
        w = ws.w[0][:k1 - k0]
        wh = ws.wh[0][:k1 - k0] if hops_t is not None else None
        dev_panel_snap(rate_t[k0:k1], n, k0, w,
                       block_next_t=next_t[k0:k1] if next_t is not None else None,
                       block_hops_t=hops_t[k0:k1] if hops_t is not None else None, w_hops_t=wh,
                       trace=trace.rows(k0, k1) if trace is not None else None)
        dev_relax_fused(rate_t, n, 0, k0, k1, w, ws, next_t=next_t, hops_t=hops_t, wh_t=wh, trace=trace,
                        updates_t=updates_t, nonneg=nonneg)
    return ws


def dev_follow_paths(next_t, src_t, dst_t, *, edge_rate_t=None, path_cap=0):
    """Batch path reconstruction on the device (fwx_dev_follow_paths).  next_t: full n x n int32
    next-hop matrix; src_t/dst_t: int32 vectors.  Returns (len, prod or None, paths or None) as
    device arrays of the same kind as next_t; asynchronous on the current / default stream."""
    n = next_t.shape[0]
    assert next_t.is_cuda and next_t.is_contiguous() and _dtype_name(next_t) == "int32"
    assert _dtype_name(src_t) == "int32" and _dtype_name(dst_t) == "int32"
    assert tuple(src_t.shape) == tuple(dst_t.shape)
    count = src_t.numel()
    len_t = _empty_like_device(next_t, (count,), "int32")
    prod_t = _empty_like_device(next_t, (count,), "float64") if edge_rate_t is not None else None
    path_t = _empty_like_device(next_t, (count, path_cap), "int32") if path_cap else None
    code = FWX_F32
    if edge_rate_t is not None:
        assert edge_rate_t.is_cuda and edge_rate_t.is_contiguous()
        assert tuple(edge_rate_t.shape) == tuple(next_t.shape)
        code = _tensor_dtype_code(edge_rate_t)
    vp = ctypes.c_void_p
    check(lib().fwx_dev_follow_paths(
        n, vp(next_t.data_ptr()), count, vp(src_t.data_ptr()), vp(dst_t.data_ptr()),
        vp(len_t.data_ptr()), vp(edge_rate_t.data_ptr()) if edge_rate_t is not None else None, code,
        vp(prod_t.data_ptr()) if prod_t is not None else None,
        vp(path_t.data_ptr()) if path_t is not None else None, path_cap, _stream_ptr(None, next_t)),
        "fwx_dev_follow_paths")
    return len_t, prod_t, path_t


def dev_solve(rate_t, *, next_t=None, hops_t=None, engine=FWX_ENGINE_AUTO, k_begin=0, k_end=0,
              serpentine=True, count_updates=False, stream=None):
    """fwx_dev_solve: whole solve (or a pivot range) on a device array holding the entire n x n
    matrix, in place, BLOCKING.  The fused engine runs with look-ahead on an internal side stream.
    Work queued on the current / default stream must be finished first: this synchronises it."""
    n = rate_t.shape[0]
    assert tuple(rate_t.shape) == (n, n)
    if stream is None:
        if _is_torch(rate_t):
            import torch
            torch.cuda.current_stream(rate_t.device).synchronize()
        else:
            from . import hip
            hip.default_stream().synchronize()
    s = _slab(rate_t, next_t, hops_t, n, 0)
    o, u = _opts(-1, engine, k_begin, k_end, 0, serpentine, count_updates, stream=stream)
    check(lib().fwx_dev_solve(ctypes.byref(s), ctypes.byref(o)), "fwx_dev_solve")
    return int(u.value) if count_updates else None
