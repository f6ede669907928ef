"""ctypes binding of the host mirror (include/fwx_host.h): the reference's operator interface for
the hot path -- buildMatrix / floydWarshall / optimum / AppState and the request layer -- written
in C++ inside libfwx.so.  Names follow /root/reference/src/lib/{Algorithms,ProcessRequests,Parsers}.hs.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import FwxError, check, lib

FWXH_ERR_ALGO = -20
FWXH_ERR_PARSE = -21
OUTSYNC, INSYNC = 0, 1

c_vp, c_i32, c_i64, c_sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
c_cp, c_dp = ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)

HOST_SIGNATURES = {
    "fwxh_session_create": (ctypes.c_int, [ctypes.POINTER(c_vp), c_i32]),
    "fwxh_session_set_devices": (ctypes.c_int, [c_vp, c_i32, ctypes.POINTER(c_i32), c_i32]),
    "fwxh_session_parts": (c_i32, [c_vp]),
    "fwxh_session_patched_solves": (c_i64, [c_vp]),
    "fwxh_session_resumed_solves": (c_i64, [c_vp]),
    "fwxh_session_resumed_pivots": (c_i64, [c_vp]),
    "fwxh_session_checkpoints_kept": (c_i32, [c_vp]),
    "fwxh_session_set_checkpoints": (ctypes.c_int, [c_vp, c_i32]),
    "fwxh_session_destroy": (ctypes.c_int, [c_vp]),
    "fwxh_session_state": (ctypes.c_int, [c_vp]),
    "fwxh_session_solves": (c_i64, [c_vp]),
    "fwxh_session_rate_count": (c_i32, [c_vp]),
    "fwxh_update_rates": (ctypes.c_int, [c_vp, c_i64, c_cp, c_cp, c_cp, ctypes.c_double,
                                         ctypes.c_double]),
    "fwxh_build_matrix": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32), c_vp, c_vp, c_cp, c_sz]),
    "fwxh_find_best_rate": (ctypes.c_int, [c_vp, c_cp, c_cp, c_cp, c_cp, c_dp, c_cp, c_sz, c_cp,
                                           c_sz]),
    "fwxh_solved_matrix": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32), c_vp, c_vp, c_vp]),
    "fwxh_optimum_dense": (ctypes.c_int, [c_i32, c_i32, ctypes.POINTER(c_cp), ctypes.POINTER(c_cp),
                                          c_vp, c_vp, c_cp, c_cp, c_cp, c_cp, c_dp, c_cp, c_sz,
                                          c_cp, c_sz]),
    "fwxh_parse_rates": (ctypes.c_int, [c_cp, ctypes.POINTER(c_i64), c_cp, c_cp, c_cp, c_sz, c_dp,
                                        c_dp, c_cp, c_sz]),
    "fwxh_parse_exch_pair": (ctypes.c_int, [c_cp, c_cp, c_cp, c_cp, c_cp, c_sz, c_cp, c_sz]),
    "fwxh_serve_line": (ctypes.c_int, [c_vp, c_cp, c_cp, c_sz]),
    "fwxh_show_double": (ctypes.c_int, [ctypes.c_double, c_cp, c_sz]),
}

_BOUND = False


def hlib():
    global _BOUND
    L = lib()
    if not _BOUND:
        for name, (res, args) in HOST_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _BOUND = True
    return L


class AlgoError(Exception):
    """Left (AlgoOptimumError text) -- Types.hs:64-67."""


class ParseError(Exception):
    """Left (ParseInputError text) -- Types.hs:59-62."""


def _b(s):
    return s.encode("utf-8")


def _vertices_from_lines(text):
    return [tuple(line.split(" ")) for line in text.split("\n") if line]


def _paren_vertices(text):
    out = []
    for line in text.split("\n"):
        if line:
            exch, ccy = line[1:-1].split(", ")
            out.append((exch, ccy))
    return out


def show_double(x):
    buf = ctypes.create_string_buffer(64)
    check(hlib().fwxh_show_double(float(x), buf, 64), "fwxh_show_double")
    return buf.value.decode()


def parse_rates(line):
    """parseRates (Parsers.hs:60-63) -> (posix, (exch,src), (exch,dst), fwd, bkd)."""
    t, fwd, bkd = c_i64(), ctypes.c_double(), ctypes.c_double()
    cap = len(line) + 8
    e, s, d = (ctypes.create_string_buffer(cap) for _ in range(3))
    err = ctypes.create_string_buffer(512 + 4 * len(line))
    rc = hlib().fwxh_parse_rates(_b(line), ctypes.byref(t), e, s, d, cap, ctypes.byref(fwd),
                                 ctypes.byref(bkd), err, len(err))
    if rc == FWXH_ERR_PARSE:
        raise ParseError(err.value.decode())
    check(rc, "fwxh_parse_rates")
    ex = e.value.decode()
    return t.value, (ex, s.value.decode()), (ex, d.value.decode()), fwd.value, bkd.value


def parse_exch_pair(line):
    """parseExchPair (Parsers.hs:65-68) -> ((exch,ccy), (exch,ccy))."""
    cap = len(line) + 8
    a, b, c, d = (ctypes.create_string_buffer(cap) for _ in range(4))
    err = ctypes.create_string_buffer(512 + 4 * len(line))
    rc = hlib().fwxh_parse_exch_pair(_b(line), a, b, c, d, cap, err, len(err))
    if rc == FWXH_ERR_PARSE:
        raise ParseError(err.value.decode())
    check(rc, "fwxh_parse_exch_pair")
    return (a.value.decode(), b.value.decode()), (c.value.decode(), d.value.decode())


def optimum_dense(vertices, rate, nxt, src, dest, n_cols=None):
    """optimum (Algorithms.hs:65-78) on dense host arrays -> (rate, start, [path vertices])."""
    n = len(vertices)
    n_cols = n if n_cols is None else n_cols
    ex = (c_cp * max(n, 1))(*[_b(v[0]) for v in vertices])
    cc = (c_cp * max(n, 1))(*[_b(v[1]) for v in vertices])
    r = ctypes.c_double()
    pbuf = ctypes.create_string_buffer(64 * (n + 2) + 256)
    ebuf = ctypes.create_string_buffer(1024)
    rp = rate.ctypes.data_as(c_vp) if rate is not None and n_cols else None
    np_ = nxt.ctypes.data_as(c_vp) if nxt is not None and n_cols else None
    rc = hlib().fwxh_optimum_dense(n, n_cols, ex, cc, rp, np_, _b(src[0]), _b(src[1]), _b(dest[0]),
                                   _b(dest[1]), ctypes.byref(r), pbuf, len(pbuf), ebuf, len(ebuf))
    if rc == FWXH_ERR_ALGO:
        raise AlgoError(ebuf.value.decode())
    check(rc, "fwxh_optimum_dense")
    vs = _paren_vertices(pbuf.value.decode())
    return r.value, vs[0], vs[1:]


class Session:
    """AppState (Types.hs:35-37) + the request layer, solves on the GPU."""

    def __init__(self, device=-1):
        h = c_vp()
        check(hlib().fwxh_session_create(ctypes.byref(h), device), "fwxh_session_create")
        self._h = h

    def set_devices(self, devices, min_vertices=0):
        """Row-partition the solved matrix over `devices` (repeats allowed) from min_vertices on."""
        arr = (c_i32 * len(devices))(*devices)
        check(hlib().fwxh_session_set_devices(self._h, len(devices), arr, int(min_vertices)),
              "fwxh_session_set_devices")

    @property
    def parts(self):
        return hlib().fwxh_session_parts(self._h)

    @property
    def patched_solves(self):
        return hlib().fwxh_session_patched_solves(self._h)

    @property
    def resumed_solves(self):
        """Patched solves that started at a checkpoint > 0 (fwx_matrix_resolve) instead of pivot 0."""
        return hlib().fwxh_session_resumed_solves(self._h)

    @property
    def resumed_pivots(self):
        return hlib().fwxh_session_resumed_pivots(self._h)

    @property
    def checkpoints_kept(self):
        """Checkpoints the resident matrix really keeps (0: it cannot resume; re-solves are full solves)."""
        return hlib().fwxh_session_checkpoints_kept(self._h)

    def set_checkpoints(self, checkpoints):
        """State checkpoints the next resident matrix keeps for resumed re-solves (0 = off)."""
        check(hlib().fwxh_session_set_checkpoints(self._h, int(checkpoints)), "fwxh_session_set_checkpoints")

    @property
    def state(self):
        return hlib().fwxh_session_state(self._h)

    @property
    def solves(self):
        return hlib().fwxh_session_solves(self._h)

    @property
    def rate_count(self):
        return hlib().fwxh_session_rate_count(self._h)

    def update_rates(self, posix, exch, src_ccy, dst_ccy, fwd, bkd):
        return bool(check(hlib().fwxh_update_rates(self._h, int(posix), _b(exch), _b(src_ccy),
                                                   _b(dst_ccy), float(fwd), float(bkd)),
                          "fwxh_update_rates"))

    def build_matrix(self):
        """buildMatrix (Algorithms.hs:26-40) -> (vertices, rate, next); no GPU."""
        n = c_i32()
        check(hlib().fwxh_build_matrix(self._h, ctypes.byref(n), None, None, None, 0),
              "fwxh_build_matrix")
        n = n.value
        rate = np.zeros((n, n), dtype=np.float64)
        nxt = np.zeros((n, n), dtype=np.int32)
        vbuf = ctypes.create_string_buffer(64 * (n + 1) + 1024)
        check(hlib().fwxh_build_matrix(self._h, ctypes.byref(c_i32()), rate.ctypes.data_as(c_vp),
                                       nxt.ctypes.data_as(c_vp), vbuf, len(vbuf)),
              "fwxh_build_matrix")
        return _vertices_from_lines(vbuf.value.decode()), rate, nxt

    def find_best_rate(self, src, dest):
        """findBestRate (ProcessRequests.hs:70-85) -> (rate, start, [path]); AlgoError on Left."""
        r = ctypes.c_double()
        n = max(self.rate_count * 2, 4)
        cap = 64 * (n + 2) + 1024
        ebuf = ctypes.create_string_buffer(1024)
        while True:
            pbuf = ctypes.create_string_buffer(cap)
            rc = hlib().fwxh_find_best_rate(self._h, _b(src[0]), _b(src[1]), _b(dest[0]),
                                            _b(dest[1]), ctypes.byref(r), pbuf, cap, ebuf, len(ebuf))
            if rc != _lib.FWX_ERR_CAPACITY or cap >= 1 << 30:
                break
            cap *= 8       # arbitrage inputs: the reference's `_path` lists revisit vertices
        if rc == FWXH_ERR_ALGO:
            raise AlgoError(ebuf.value.decode())
        if rc < 0:
            raise FwxError(rc, "fwxh_find_best_rate: " + ebuf.value.decode())
        vs = _paren_vertices(pbuf.value.decode())
        return r.value, vs[0], vs[1:]

    def solved_matrix(self):
        """floydWarshall of the current rates (Algorithms.hs:19-20) -> (rate, next, hops)."""
        n = c_i32()
        check(hlib().fwxh_build_matrix(self._h, ctypes.byref(n), None, None, None, 0), "n")
        n = n.value
        rate = np.zeros((n, n), dtype=np.float64)
        nxt = np.zeros((n, n), dtype=np.int32)
        hops = np.zeros((n, n), dtype=np.int32)
        check(hlib().fwxh_solved_matrix(self._h, ctypes.byref(c_i32()), rate.ctypes.data_as(c_vp),
                                        nxt.ctypes.data_as(c_vp), hops.ctypes.data_as(c_vp)),
              "fwxh_solved_matrix")
        return rate, nxt, hops

    def serve_line(self, line):
        """One turn of Main.userPrompt (Main.hs:18-37): the printed lines."""
        cap = 1 << 16
        while True:
            buf = ctypes.create_string_buffer(cap)
            rc = hlib().fwxh_serve_line(self._h, _b(line), buf, cap)
            if rc == _lib.FWX_ERR_CAPACITY:
                cap *= 4
                continue
            check(rc, "fwxh_serve_line")
            text = buf.value.decode()
            assert text.endswith("\n")
            return text[:-1].split("\n")

    def close(self):
        if self._h:
            hlib().fwxh_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
