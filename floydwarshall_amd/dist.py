"""Row-block partitioned solve: one process per GPU, torch.distributed (RCCL over xGMI).

The reference is single-threaded; this is the multi-GPU shape of the same
loop.  Rank r owns the contiguous rows [bounds[r], bounds[r+1]) of `rate`
(and of `next`).  Step k of runAlgo (/root/reference/src/lib/Algorithms.hs:42-61) needs, for a
local row i, only r[i][k] / next[i][k] (local) and the pivot row r[k][.] AS IT STANDS AT THE START
OF STEP k, which lives on one rank.  That is the only exchange, so the only collective is a
broadcast of pivot rows.

Latency, not bandwidth, is the problem (N messages of N*4 bytes): pivots are therefore exchanged
B at a time.  The owner of pivots [k0, k0+B) runs the PANEL phase -- it evolves a scratch copy of
just those B rows through those B pivots in exact k order and exports the time-k snapshot of each
pivot row -- and broadcasts the B x N snapshot panel once; every rank then applies the B pivots to
ALL its rows (the pivot rows included) using the snapshots.  No operand and no order changes, so the result is bit-identical to the
single-GPU solve (tests/test_dist_gloo.py, tests/test_gpu_parity.py).

Look-ahead: while a rank relaxes its slab with panel b, the owner of panel b+1 first brings just
those B rows up to date, runs their panel phase and starts the broadcast, so the exchange of
panel b+1 overlaps the bulk of step b.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import engine


def _lib_unsupported():
    from ._lib import FWX_ERR_UNSUPPORTED
    return FWX_ERR_UNSUPPORTED


def row_bounds(n, world):
    """Contiguous, balanced row blocks: rank r owns [bounds[r], bounds[r+1])."""
    return [(n * r) // world for r in range(world + 1)]


def pivot_blocks(n, world, block):
    """[(k0, B, owner)]: pivot panels never straddle two owners."""
    bounds = row_bounds(n, world)
    out = []
    for r in range(world):
        k0 = bounds[r]
        while k0 < bounds[r + 1]:
            b = min(block, bounds[r + 1] - k0)
            out.append((k0, b, r))
            k0 += b
    return out


class Slab:
    """The arrays one rank holds for its rows: rate, optional next / hops (torch tensors, rows x n)
    and an optional engine.Trace.  rows(lo, hi) slices all of them alike."""

    def __init__(self, rate, nxt=None, hops=None, trace=None):
        self.rate, self.nxt, self.hops, self.trace = rate, nxt, hops, trace

    @property
    def nrows(self):
        return self.rate.shape[0]

    def rows(self, lo, hi):
        cut = lambda t: None if t is None else t[lo:hi]  # noqa: E731
        return Slab(self.rate[lo:hi], cut(self.nxt), cut(self.hops),
                    None if self.trace is None else self.trace.rows(lo, hi))


class HipBackend:
    """The product backend: libfwx kernels on torch-owned device memory, current stream.

    engine "fused": fwx_dev_relax_fused (64 pivots per pass; carries next, hops and the path
    trace); "perk": fwx_dev_relax (one launch per pivot; next and hops, no trace on slabs).  Both
    consume the same snapshot panels and give the same bits."""

    def __init__(self, engine_name="fused"):
        assert engine_name in ("fused", "perk")
        self.engine_name = engine_name
        self.ws = None
        self.nonneg = False

    def check_domain(self, slab, n, row0):
        """Domain bits of this slab (fwx.h "Domain"): bit 0 = every rate >= +0 and not NaN, bit 1 =
        no positive rate without a path.  solve_partitioned combines the answers of all ranks (the
        domain must hold globally)."""
        if slab.nrows == 0:
            return 3
        return engine.dev_domain_bits(slab.rate, n, row0, slab.nxt)

    def prepare(self, n, rows, dtype, device, with_next, with_hops=False):
        self.ws = engine.FusedWorkspace(n, rows, dtype, device, with_next=with_next, with_hops=with_hops)

    def panel(self, block, n, k0, w, wh=None):
        """Snapshot panel of the pivot rows `block` (the matrix is not modified); wh: their hops."""
        engine.dev_panel_snap(block.rate, n, k0, w, block_next_t=block.nxt, block_hops_t=block.hops,
                              w_hops_t=wh, trace=block.trace)

    def _fused(self, slab, n, row0, k0, k1, w, wh, skip=None):
        engine.dev_relax_fused(slab.rate, n, row0, k0, k1, w, self.ws, next_t=slab.nxt, hops_t=slab.hops,
                               wh_t=wh, trace=slab.trace, nonneg=self.nonneg, skip=skip)

    def _perk(self, slab, n, row0, k0, k1, w, wh, skip=None):
        assert slab.trace is None, "the per-k engine keeps no path trace on slabs: use the fused engine"
        engine.dev_relax(slab.rate, n, row0, k0, k1, pivots_t=w, pivot_hops_t=wh, next_t=slab.nxt,
                         hops_t=slab.hops, skip=skip)

    def relax(self, slab, n, row0, k0, k1, w, wh=None):
        if slab.nrows == 0:
            return
        (self._fused if self.engine_name == "fused" else self._perk)(slab, n, row0, k0, k1, w, wh)

    def relax_skipping(self, slab, n, row0, k0, k1, w, wh, skip):
        """The whole slab except the rows skip = (lo, hi) in ONE launch per pivot; False if this
        backend / alignment cannot do it (the caller then relaxes above and below separately)."""
        lo, hi = skip
        if self.engine_name == "fused":
            if lo % 8 or hi % 8:
                return False
            self._fused(slab, n, row0, k0, k1, w, wh, skip=skip)
            return True
        if lo % 4 or hi % 4:
            return False
        self._perk(slab, n, row0, k0, k1, w, wh, skip=skip)
        return True

    def relax_lookahead(self, slab, n, row0, k0, k1, w, wh=None):
        """The few rows the next panel is made of.  They sit on the owner's critical path, so
        they always take the one-launch fused kernel (bit-identical to 64 per-k launches)."""
        if slab.nrows == 0:
            return
        if n % (16 // slab.rate.element_size()) == 0:
            self._fused(slab, n, row0, k0, k1, w, wh)
        else:
            self.relax(slab, n, row0, k0, k1, w, wh)


class StepTimer:
    """Per-step spans of one rank's solve_partitioned, from timing events on the stream the work runs on
    (GPU) or the wall clock (CPU rehearsal): bulk = the slab sweep of a step, lookahead = the owner bringing
    the next block's rows up to date, panel = the owner's snapshot panel, exchange = from the broadcast's
    issue to the side stream having waited for it.  A diagnostic (two event records per span): the
    benchmark runs it in an extra, untimed solve.  summary() -> {kind: (mean microseconds, count)}."""

    KINDS = ("bulk", "lookahead", "panel", "exchange")

    def __init__(self, on_gpu):
        self.on_gpu = on_gpu
        self.spans = []           # (kind, e0, e1) or (kind, seconds)

    def begin(self, kind):
        if self.on_gpu:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            return (kind, e0)
        import time
        return (kind, time.perf_counter())

    def end(self, token):
        kind, start = token
        if self.on_gpu:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.spans.append((kind, start, e1))
        else:
            import time
            self.spans.append((kind, time.perf_counter() - start))

    def summary(self):
        if self.on_gpu:
            torch.cuda.synchronize()
        acc = {k: [0.0, 0] for k in self.KINDS}
        for sp in self.spans:
            us = 1e3 * sp[1].elapsed_time(sp[2]) if self.on_gpu else 1e6 * sp[1]
            acc[sp[0]][0] += us
            acc[sp[0]][1] += 1
        return {k: ((v[0] / v[1]) if v[1] else 0.0, v[1]) for k, v in acc.items()}


def solve_partitioned(rate, n, rank, world, *, nxt=None, hops=None, trace=None, block=64, backend=None,
                      group=None, lookahead=True, force_collectives=False, skip_launch=True, timer=None):
    """In-place solve of this rank's slab `rate` (rows row_bounds(n, world)[rank]...), with its
    next-hops `nxt`, path lengths `hops` (needs nxt) and path trace `trace` (an engine.Trace of the
    slab, all -1; needs nxt) if given.

    All ranks must call this with the same n / world / block / lookahead.  Works on any device the
    backend and the process group support (HIP + RCCL in production; the tests drive it on
    CPU + gloo with an oracle-backed backend to check the schedule).

    Schedule per panel b (pivots [k0, k0+B), snapshots W_b; with hops also WH_b, the hops of the
    pivot rows at the same times, which travel with them).  A panel is a pure SNAPSHOT of the
    pivot rows (they are not modified by it), so every rank relaxes ALL its rows with W_b:
        wait for W_b
        owner of panel b+1: relax ONLY the rows of panel b+1 with W_b, then snapshot them
        everyone:           start the broadcast of W_{b+1} (async, other buffer)
        everyone:           relax the rest of the slab with W_b   <- overlaps the broadcast
    skip_launch: the owner relaxes "the rest" in one launch per pivot that leaves the look-ahead
    rows alone (backend.relax_skipping) instead of one launch above and one below them.
    """
    backend = backend or HipBackend()
    collectives = world > 1 or force_collectives   # force: rehearse the RCCL calls on one rank
    assert 1 <= block <= engine.FWX_FUSED_BLOCK
    assert (hops is None and trace is None) or nxt is not None
    bounds = row_bounds(n, world)
    row0, rows = bounds[rank], bounds[rank + 1] - bounds[rank]
    assert tuple(rate.shape) == (rows, n)
    slab = Slab(rate, nxt, hops, trace)
    blocks = pivot_blocks(n, world, block)
    if not blocks:
        return
    if hasattr(backend, "prepare"):
        backend.prepare(n, rows, rate.dtype, rate.device, nxt is not None, hops is not None)
    if hasattr(backend, "check_domain"):
        bits = backend.check_domain(slab, n, row0)
        ok = torch.tensor([bits & 1, (bits >> 1) & 1], dtype=torch.int32, device=rate.device)
        if collectives:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        d1, d2 = (int(v) for v in ok.tolist())
        if nxt is not None and not (d1 and d2):
            # The slab kernels take next[i][k] as the head of ikPath ++ kjPath (Algorithms.hs:55);
            # outside the reference's domain that needs the pivot rows' next-hops as well, which
            # the snapshot panels do not carry: such a matrix is solved on one GPU (fwx_solve_*).
            raise engine.FwxError(_lib_unsupported(), "solve_partitioned: matrix outside the "
                                  "reference's domain (negative/NaN rate or a positive rate without "
                                  "a path) with next-hops")
        # max-form kernels: fused engine, every entry >= +0 and not NaN on every rank (and D2 with next)
        backend.nonneg = bool(d1) and (nxt is None or bool(d2)) and \
            getattr(backend, "engine_name", "") == "fused"
    bufs = [torch.empty((block, n), dtype=rate.dtype, device=rate.device) for _ in range(2)]
    hbufs = [torch.empty((block, n), dtype=torch.int32, device=rate.device) for _ in range(2)] \
        if hops is not None else [None, None]
    # On a GPU the panel phase and the broadcast run on a SIDE stream, so that on the owner they
    # overlap the bulk relax of the previous panel instead of queueing behind it.
    on_gpu = rate.is_cuda
    main = torch.cuda.current_stream(rate.device) if on_gpu else None
    side = torch.cuda.Stream(device=rate.device) if on_gpu else None

    def panel_and_broadcast(idx):
        """Snapshot panel idx (owner) + its broadcast; returns (w, wh, wait) where wait() orders
        the current stream behind both."""
        k0, b, owner = blocks[idx]
        w = bufs[idx & 1][:b]
        wh = hbufs[idx & 1][:b] if hops is not None else None
        works = []

        tok = []

        def issue():
            if rank == owner:
                lo = k0 - row0
                t = timer.begin("panel") if timer else None
                backend.panel(slab.rows(lo, lo + b), n, k0, w, wh)
                if timer:
                    timer.end(t)
            if collectives:
                src = owner if group is None else dist.get_global_rank(group, owner)
                if timer:
                    tok.append(timer.begin("exchange"))
                works.append(dist.broadcast(w, src=src, group=group, async_op=True))
                if wh is not None:
                    works.append(dist.broadcast(wh, src=src, group=group, async_op=True))

        if on_gpu:
            side.wait_stream(main)                 # everything queued so far precedes the panel
            with torch.cuda.stream(side):
                issue()
        else:
            issue()

        def wait():
            for work in works:
                if on_gpu:
                    with torch.cuda.stream(side):
                        work.wait()                # side stream behind the collective
                else:
                    work.wait()
            if tok:
                if on_gpu:
                    with torch.cuda.stream(side):
                        timer.end(tok.pop())
                else:
                    timer.end(tok.pop())
            if on_gpu:
                main.wait_stream(side)
        return w, wh, wait

    def relax_rows_except(skips, k0, k1, w, wh):
        t = timer.begin("bulk") if timer else None
        done = skip_launch and len(skips) == 1 and hasattr(backend, "relax_skipping") and \
            backend.relax_skipping(slab, n, row0, k0, k1, w, wh, skips[0])
        pos = 0
        for lo, hi in ([] if done else sorted(skips) + [(rows, rows)]):
            if lo > pos:
                backend.relax(slab.rows(pos, lo), n, row0 + pos, k0, k1, w, wh)
            pos = max(pos, hi)
        if timer:
            timer.end(t)

    w, wh, wait = panel_and_broadcast(0)
    for idx, (k0, b, owner) in enumerate(blocks):
        wait()
        k1 = k0 + b
        skips = []
        nxt_panel = None
        if idx + 1 < len(blocks) and lookahead:
            nk0, nb, nowner = blocks[idx + 1]
            if rank == nowner:
                nlo = nk0 - row0
                t = timer.begin("lookahead") if timer else None
                getattr(backend, "relax_lookahead", backend.relax)(
                    slab.rows(nlo, nlo + nb), n, nk0, k0, k1, w, wh)
                if timer:
                    timer.end(t)
                skips.append((nlo, nlo + nb))
            nxt_panel = panel_and_broadcast(idx + 1)
            relax_rows_except(skips, k0, k1, w, wh)
        else:
            relax_rows_except(skips, k0, k1, w, wh)
            if idx + 1 < len(blocks):
                nxt_panel = panel_and_broadcast(idx + 1)
        if nxt_panel is not None:
            w, wh, wait = nxt_panel
    if on_gpu:
        main.wait_stream(side)


# ------------------------------------------------------------------------------------------------------
# One partition per process behind the C ABI: fwx_matrix_create_part + a torch.distributed exchange
# ------------------------------------------------------------------------------------------------------
class _DevPtr:
    """A device buffer libfwx owns, seen by torch through the CUDA array interface (no copy)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


class PartMatrix(engine.DeviceMatrix):
    """Partition `rank` of `world` row blocks as an fwx_matrix in THIS process (fwx_matrix_create_part): the
    schedules, kernels, path trace, kept input and resume are libfwx's own -- the very code the one-process
    partitioned handle runs, incl. the 128-pivot pair schedule --, and the one exchange of the algorithm, the
    snapshot panel of 64 pivot rows, is a `torch.distributed.broadcast` issued from libfwx's callback on the
    partition's side stream (RCCL over xGMI in production; gloo for rehearsals with several ranks on one GPU).

    upload / download work on this rank's row block: rows [row0, row0 + rows) x n, as the LIBRARY places them
    (self.row0 / self.rows, bounds(): 64-aligned for n >= 128 * world; not row_bounds()).  solve() first
    combines the ranks' domain bits (an all-reduce), as the C header asks; count_updates returns this rank's
    share of U.  Every rank must make the same calls with the same options."""

    def __init__(self, n, dtype, rank, world, with_next=True, with_hops=False, device=-1, group=None):
        from ._lib import EXCHANGE_FN
        self.n, self.dtype = int(n), np.dtype(dtype)
        self.with_next, self.with_hops = bool(with_next), bool(with_hops)
        self.rank, self.world, self.group = int(rank), int(world), group
        self._views = {}
        self._error = None
        self._cb = EXCHANGE_FN(self._exchange)            # (kept alive with the handle)
        h = ctypes.c_void_p()
        code = engine.FWX_F64 if self.dtype == np.float64 else engine.FWX_F32
        engine.check(engine.lib().fwx_matrix_create_part(ctypes.byref(h), self.n, code, int(self.with_next),
                                                         int(self.with_hops), self.rank, self.world, device,
                                                         ctypes.cast(self._cb, ctypes.c_void_p), None),
                     "fwx_matrix_create_part")
        self._h = h
        self._domain_fresh = False
        # the library places the partitions (64-aligned for n >= 128 * world, so that the pair schedule and
        # resume checkpoints apply to any n): ask it, do not assume row_bounds()
        self.row0, self.rows = self.part_rows(self.rank)

    # -- the exchange: called by libfwx on the solving thread, once per panel, same order on every rank --
    def _view(self, ptr, count, typestr, torch_dtype):
        key = (int(ptr), int(count), typestr)
        t = self._views.get(key)
        if t is None:
            t = torch.as_tensor(_DevPtr(ptr, count, typestr), device="cuda")
            assert t.dtype == torch_dtype and t.data_ptr() == int(ptr)
            if len(self._views) > 4096:
                self._views.clear()
            self._views[key] = t
        return t

    def _exchange(self, ctx, k0, bt, owner, w, wh, count, stream):
        try:
            src = owner if self.group is None else dist.get_global_rank(self.group, owner)
            ts = "<f8" if self.dtype == np.float64 else "<f4"
            tdt = torch.float64 if self.dtype == np.float64 else torch.float32
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))):
                work = [dist.broadcast(self._view(w, count, ts, tdt), src=src, group=self.group, async_op=True)]
                if wh:
                    work.append(dist.broadcast(self._view(wh, count, "<i4", torch.int32), src=src, group=self.group,
                                               async_op=True))
                for x in work:
                    x.wait()                      # the side stream waits for the collective, not the host
            return 0
        except Exception as err:                  # never let an exception unwind into C
            self._error = err
            return -1

    def bounds(self):
        """[first row of partition 0, ..., of partition world - 1, n]: where the library cut the matrix."""
        return [self.part_rows(p)[0] for p in range(self.world)] + [self.n]

    # -- slab-shaped transfers ------------------------------------------------------------------------------
    def upload(self, rate, nxt=None, hops=None):
        for a in (rate, nxt, hops):
            assert a is None or (a.shape == (self.rows, self.n) and a.flags.c_contiguous)
        assert rate.dtype == self.dtype
        engine.check(engine.lib().fwx_matrix_upload(self._h, rate.ctypes.data_as(ctypes.c_void_p),
                                                    None if nxt is None else nxt.ctypes.data_as(ctypes.c_void_p),
                                                    None if hops is None else hops.ctypes.data_as(ctypes.c_void_p)),
                     "fwx_matrix_upload")
        self._domain_fresh = False

    def upload_dev(self, rate_d, next_d=None, hops_d=None):
        super().upload_dev(rate_d, next_d, hops_d)
        self._domain_fresh = False

    def download(self):
        rate = np.empty((self.rows, self.n), dtype=self.dtype)
        nxt = np.empty((self.rows, self.n), dtype=np.int32) if self.with_next else None
        hops = np.empty((self.rows, self.n), dtype=np.int32) if self.with_hops else None
        ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        engine.check(engine.lib().fwx_matrix_download(self._h, ptr(rate), ptr(nxt), ptr(hops)), "fwx_matrix_download")
        return rate, nxt, hops

    def patch_input(self, index, rate_vals, next_vals=None, hops_vals=None):
        super().patch_input(index, rate_vals, next_vals, hops_vals)
        self._domain_fresh = False

    def _vote(self):
        """The domain must hold on EVERY rank (fwx.h "Domain"): local bits -> all-reduce AND -> set."""
        if self._domain_fresh:
            return
        bits = ctypes.c_int32(3)
        engine.check(engine.lib().fwx_matrix_domain_bits(self._h, ctypes.byref(bits)), "fwx_matrix_domain_bits")
        ok = torch.tensor([bits.value & 1, (bits.value >> 1) & 1], dtype=torch.int32, device="cuda")
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        d1, d2 = (int(v) for v in ok.tolist())
        engine.check(engine.lib().fwx_matrix_set_domain(self._h, d1 | (d2 << 1)), "fwx_matrix_set_domain")
        self._domain_fresh = True

    def solve(self, **kw):
        self._vote()
        self._error = None
        try:
            return super().solve(**kw)
        except engine.FwxError:
            if self._error is not None:
                raise self._error
            raise

    def resolve(self, index, rate_vals, next_vals=None, hops_vals=None, **kw):
        self._error = None
        try:
            return super().resolve(index, rate_vals, next_vals, hops_vals, **kw)
        except engine.FwxError:
            if self._error is not None:
                raise self._error
            raise
