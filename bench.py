#!/usr/bin/env python3
"""bench.py -- edge-relaxations/s (N^3/t) of the max-product Floyd-Warshall hot path on MI355X.

Contract (one JSON line from rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 runs either way the driver may start it:
    python bench.py --gpus N ...                     ONE process, N devices: the row-partitioned handle
        behind the C ABI (fwx_matrix_create_multi, panels on RCCL) -- what a reference host binds;
        torch-free like N = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        one process per GPU (WORLD_SIZE set): floydwarshall_amd.dist over torch.distributed / RCCL
  --devices 0,0,...  lists the partitions' devices explicitly; a repeated device makes LOGICAL
        partitions on one GPU (rehearses the whole N > 1 path where only one GPU exists; the line is
        flagged INVALID_logical_partitions and is never a performance number)

A "step" is ONE FULL SOLVE (all N pivots of runAlgo, /root/reference/src/lib/Algorithms.hs:42-61)
of the synthetic dense matrix, restarted from the pristine input (a device-to-device copy inside
the timed region, ~0.03 % of a step).  Workload = BASELINE.json's metric configuration:
N = 16384, fp32, dense D1 input (floydwarshall_amd/synth.py), matrices resident in HBM.

  value     = K * N^3 / t      t = wall time of the K steps, barrier + device synchronize on both
                               sides, max over ranks
  roofline  = per-k kernel `relax_k` against HBM: algorithmic bytes per launch
              (s*N^2 + s*U/N + 2*s*N) / average launch duration measured
              live with HIP events on the launch stream over the timed region
  cpu_baseline = the oracle's multithreaded dense loop (a C restatement of the reference loop --
              the Haskell reference cannot be built here) on a bounded k-slice of the same matrix

The single-process forms run WITHOUT torch: device buffers, the stream and the events come from the
HIP runtime directly (floydwarshall_amd/hip.py), so the process holds one HIP runtime -- the one
libfwx is built against -- and `rocprofv3 --pmc ... -- python3 bench.py ...` profiles exactly the
benchmarked launches.  Only the torchrun form imports torch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import threading

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC; RCCL needs it before HIP initialises
# (floydwarshall_amd/_lib.py sets the same default for every other binding of the package)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SEGMENTS = 16
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
INFINITY_CACHE_BYTES = 256 << 20
PMC_PROFILE = os.path.join("profiles", "r04_pmc_traffic.json")


class Watchdog:
    """A hung collective must not end as a driver kill with nothing written.  arm(label) starts the
    clock for one phase (a step, the handle creation, an extra leg); if the phase is still running
    `bound_s` seconds later the watchdog thread prints ONE JSON line that says which phase hung, flushes,
    and ends the process with exit code 3 (os._exit: the main thread is stuck inside a blocking runtime
    call and cannot be unwound; nothing is re-executed).  Only `emit`-ing ranks print (rank 0)."""

    EXIT_CODE = 3

    def __init__(self, bound_s, base_line, emit=True, out=None, exit_fn=None):
        self.bound_s, self.base, self.emit = float(bound_s), dict(base_line), emit
        self.out = out or sys.stdout
        self.exit_fn = exit_fn or os._exit
        self._lock = threading.Lock()
        self._label, self._deadline, self._done = None, None, []
        self._valid = None
        self._quiet = False
        self._stop = threading.Event()
        self._thread = None
        if self.bound_s > 0:
            self._thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
            self._thread.start()

    def arm(self, label):
        with self._lock:
            self._label, self._deadline = label, time.monotonic() + self.bound_s

    def disarm(self):
        with self._lock:
            if self._label is not None:
                self._done.append(self._label)
            self._label = self._deadline = None

    def stop(self):
        self._stop.set()

    def set_valid_line(self, line):
        """From here on the run HAS its result: `line` (the dict the caller keeps filling) carries `value`
        from the completed timed region.  If an optional leg hangs after this point the watchdog prints that
        line as it stands, plus `extras_aborted`, and exits with code 0 -- a diagnostic leg must not cost the
        measurement."""
        with self._lock:
            self._valid = line

    def line_printed(self):
        """The line is out.  A phase that hangs from here on (the final barrier with a rank gone) only ends the
        process, with code 0 and nothing more written."""
        with self._lock:
            self._quiet = True

    def error_line(self, label):
        line = dict(self.base)
        line.update({"value": None, "error": "watchdog: phase %r still running after %.0f s" % (label, self.bound_s),
                     "hung_phase": label, "phases_completed": list(self._done)})
        return line

    def _run(self):
        while not self._stop.wait(0.25):
            with self._lock:
                label, deadline = self._label, self._deadline
            if label is not None and time.monotonic() > deadline:
                with self._lock:
                    valid, quiet = self._valid, self._quiet
                if quiet:
                    self.exit_fn(0)
                    return
                if valid is not None:
                    line = dict(valid)
                    line["extras_aborted"] = {"hung_phase": label, "after_s": self.bound_s,
                                              "phases_completed": list(self._done)}
                    try:
                        text = json.dumps(line)
                    except (TypeError, ValueError, RuntimeError):
                        text = json.dumps({k: v for k, v in line.items() if isinstance(v, (int, float, str, bool, type(None)))})
                    if self.emit:
                        print(text, file=self.out, flush=True)
                    self.exit_fn(0)
                    return
                if self.emit:
                    print(json.dumps(self.error_line(label)), file=self.out, flush=True)
                self.exit_fn(self.EXIT_CODE)
                return


def create_multi_handle(engine, n, np_dtype, with_next, devs, exchange_name, factory=None):
    """The partitioned handle of the single-process N > 1 form, fail-soft: with `--exchange auto` a
    handle that cannot be created over RCCL (FWX_ERR_RCCL: librccl missing, ncclCommInitAll failing -- e.g.
    without dmabuf IPC) is created again over peer copies, which need neither RCCL nor IPC inside one
    process, and the line says so.  Returns (handle, {"requested", "rccl_error"})."""
    factory = factory or engine.DeviceMatrix
    xchg = {"auto": engine.FWX_XCHG_AUTO, "rccl": engine.FWX_XCHG_RCCL, "peer": engine.FWX_XCHG_PEER}[exchange_name]
    info = {"requested": exchange_name, "rccl_error": None}
    try:
        return factory(n, np_dtype, with_next=with_next, devices=devs, exchange=xchg), info
    except engine.FwxError as err:
        if exchange_name != "auto" or err.status != engine.FWX_ERR_RCCL:
            raise
        info["rccl_error"] = {"status": int(err.status), "message": str(err), "where": "fwx_matrix_create_multi"}
    return factory(n, np_dtype, with_next=with_next, devices=devs, exchange=engine.FWX_XCHG_PEER), info


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", dest="n", type=int, default=16384, help="matrix order N")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--dist", default="d1", choices=["d1", "d2"])
    ap.add_argument("--with-next", action="store_true", help="carry the next-hop matrix")
    ap.add_argument("--engine", default="perk", choices=["perk", "fused"],
                    help="perk: one launch per pivot (the HBM-roofline kernel the metric is defined "
                         "on); fused: 64 pivots per pass (VALU-bound, same bits)")
    ap.add_argument("--block", type=int, default=64, help="pivots per broadcast (N > 1, torchrun form)")
    ap.add_argument("--devices", default="",
                    help="comma-separated HIP ordinals, one per row partition (single-process N > 1 "
                         "form); default 0..N-1.  A repeated ordinal = logical partitions (rehearsal)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "peer"],
                    help="panel transport of the single-process N > 1 form")
    ap.add_argument("--no-serpentine", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region: no fused / f64 / serpentine-off / cross-check legs")
    ap.add_argument("--no-fused-extra", action="store_true")
    ap.add_argument("--no-f64-extra", action="store_true")
    ap.add_argument("--driver", default="part", choices=["part", "legacy"],
                    help="torchrun form: part = one partition per process behind the C ABI (libfwx's own schedule, "
                         "panels broadcast by torch.distributed from its callback); legacy = dist.solve_partitioned")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (gloo only to rehearse the N > 1 code path "
                         "with several ranks on one GPU; never a performance number)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--step-timeout", type=float, default=240.0,
                    help="N > 1: seconds one phase (handle creation, a step, an extra leg) may take before the "
                         "watchdog prints an error line and exits with code 3; 0 = off")
    ap.add_argument("--kslice", type=int, default=0,
                    help="DEBUG: run only this many pivots per step (result flagged invalid)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="preset for one of BASELINE.json's configs (2: N=1024 f64 per-k; 3: N=8192 f32 "
                         "fused; 4: the default headline; 5: N=32768 f32 + next-hop matrix, fused)")
    args = ap.parse_args()
    if args.config == 2:
        args.n, args.dtype, args.engine = 1024, "f64", "perk"
    elif args.config == 3:
        args.n, args.dtype, args.engine = 8192, "f32", "fused"
    elif args.config == 5:
        args.n, args.dtype, args.engine, args.with_next = 32768, "f32", "fused", True
    if args.no_extras:
        args.no_fused_extra = args.no_f64_extra = True
    args.device_list = parse_devices(args.devices, args.gpus)
    if args.devices:
        args.gpus = len(args.device_list)
    return args


def parse_devices(text, gpus):
    """--devices "0,0,1" -> [0, 0, 1]; empty -> [0, ..., gpus-1].  Raises SystemExit on nonsense."""
    if not text:
        if gpus < 1:
            raise SystemExit("--gpus must be >= 1")
        return list(range(gpus))
    try:
        devs = [int(x) for x in text.split(",")]
    except ValueError:
        raise SystemExit("--devices wants comma-separated device ordinals, got %r" % text)
    if not devs or min(devs) < 0 or len(devs) > 32:
        raise SystemExit("--devices: 1..32 non-negative ordinals, got %r" % text)
    return devs


def host_cores():
    """Threads this process may actually run: CPU affinity capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(rate_host, cpu_seconds):
    """Oracle (port of the reference loop) on a bounded k-slice, all host cores of this process."""
    import oracle
    n = rate_host.shape[0]
    cores = host_cores()
    work = rate_host.copy()
    done, t_total, chunk = 0, 0.0, 8
    while t_total < cpu_seconds and done < n:
        k1 = min(n, done + chunk)
        t0 = time.perf_counter()
        oracle.relax_mt(work, None, done, k1, threads=cores)
        t_total += time.perf_counter() - t0
        done = k1
    relax = float(done) * n * n
    # the same loop on ONE thread (SURVEY.md section 8d asks for both), a few pivots further on
    st_done, st_total = done, 0.0
    while st_total < cpu_seconds / 4 and st_done < n:
        t0 = time.perf_counter()
        oracle.relax(work, None, None, st_done, st_done + 1)
        st_total += time.perf_counter() - t0
        st_done += 1
    single = {"value": float(st_done - done) * n * n / st_total if st_total > 0 else None,
              "cores": 1, "sample": "pivots [%d,%d) (%.1f s)" % (done, st_done, st_total)}
    return {"value": relax / t_total, "unit": "edge-relaxations/s", "cores": cores,
            "single_thread": single, "kind": "port",
            "sample": "pivots [0,%d) of the same N=%d %s matrix (%.1f s); oracle/fw_oracle.c "
                      "fwo_relax_mt, a C restatement of Algorithms.hs:42-61 -- the Haskell "
                      "reference cannot be built in this image" % (done, n, rate_host.dtype, t_total)}


def pmc_traffic(n, args):
    """HBM bytes per relax_k launch from the rocprofv3 PMC passes over THIS command (`rocprofv3 --pmc
    FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline`, then
    WRITE_SIZE; tools/pmc_summary.py turns the counter CSVs into profiles/r04_pmc_traffic.json).
    Counters cannot be read from inside the process, so the line quotes the committed summary, and
    only for the exact configuration it was measured on; otherwise null."""
    path = os.path.join(ROOT, PMC_PROFILE)
    if n != 16384 or args.dtype != "f32" or args.with_next or args.kslice or args.engine != "perk" \
            or not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    return t["traffic_bytes_per_launch"], t.get("source", PMC_PROFILE)


def digest(a):
    """64-bit digest of an array's bytes (xxhash if present, else crc32 of 64 MiB chunks); a list / tuple of
    arrays is digested as their concatenation (row slabs of one matrix, in order)."""
    parts = a if isinstance(a, (list, tuple)) else [a]
    bufs = [memoryview(np.ascontiguousarray(x)).cast("B") for x in parts]
    try:
        import xxhash
        h = xxhash.xxh64()
        for buf in bufs:
            for off in range(0, len(buf), 1 << 26):
                h.update(buf[off:off + (1 << 26)])
        return "xxh64:" + h.hexdigest()
    except ImportError:
        import zlib
        c = 0
        for buf in bufs:
            for off in range(0, len(buf), 1 << 26):
                c = zlib.crc32(buf[off:off + (1 << 26)], c)
        return "crc32:%08x" % c


def alg_bytes_per_launch(n, es, u_per_launch, with_next):
    # B_alg = s*N^3 + s*U + 2*s*N^2 per solve (+4*U with next), per launch:
    return es * n * n + (es + (4 if with_next else 0)) * u_per_launch + 2 * es * n


def workload_config(args, n, k_end, serp, world, partition=None):
    return {"workload": "N=%d dense %s rate matrix (%s), full solve = %d pivot steps per step, %s%s"
                        % (n, args.dtype, args.dist.upper(), k_end,
                           "per-k engine" if args.engine == "perk"
                           else "fused engine (64 pivots per pass)",
                           ", with next-hop matrix" if args.with_next else ""),
            "n": n, "engine": args.engine, "serpentine": serp,
            "partition": partition or ("single GPU" if world == 1 else
                                       "row-block x%d, %d-pivot snapshot panels broadcast on %s, one process "
                                       "per GPU (torch.distributed)"
                                       % (world, args.block, "RCCL" if args.backend == "nccl" else "gloo (rehearsal)"))}


def make_input(args, n):
    """D1/D2 in float64 (the generators draw in f64 and round): (rate64, next)."""
    from floydwarshall_amd import synth
    cfg_index = {0: 3, 2: 1, 3: 2, 4: 3, 5: 4}[args.config]  # seed = BASE_SEED + configs[] index
    return synth.GENERATORS[args.dist](n, np.float64, synth.BASE_SEED + cfg_index)


# ------------------------------------------------------------------------------------------------
# N = 1: torch-free
# ------------------------------------------------------------------------------------------------
def perk_solve(engine, hip, rate, nxt, n, k_end, serp, stream, upd=None, timed=False):
    """All pivots of one solve as back-to-back launches on `stream`; with timed=True the pivots are
    issued in SEGMENTS groups with a HIP event between groups (no synchronisation): per-segment
    launch time shows how the cost moves with k.  Returns (events, segment bounds)."""
    segs = [k_end * i // SEGMENTS for i in range(SEGMENTS + 1)] if timed else [0, k_end]
    evs = []
    for a, b in zip(segs[:-1], segs[1:]):
        if timed:
            evs.append(hip.Event())
            evs[-1].record(stream)
        if b > a:
            engine.dev_relax(rate, n, 0, a, b, next_t=nxt, serpentine=serp, updates_t=upd,
                             stream=stream)
    if timed:
        evs.append(hip.Event())
        evs[-1].record(stream)
    return evs, segs


def run_single(args):
    from floydwarshall_amd import engine, hip
    assert "torch" not in sys.modules, "the N=1 benchmark must stay torch-free"
    n = args.n
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    es = np.dtype(np_dtype).itemsize
    rate64, next_host = make_input(args, n)
    rate_host = rate64 if np_dtype == np.float64 else rate64.astype(np.float32)
    want_f64 = (np_dtype == np.float32 and args.engine == "perk" and not args.kslice
                and not args.no_f64_extra and not args.with_next)
    if not want_f64:
        del rate64

    hip.set_device(0)
    stream = hip.Stream()
    pristine = hip.DeviceArray.from_numpy(rate_host)
    rate = hip.DeviceArray(pristine.shape, np_dtype)
    pristine_next = nxt = None
    if args.with_next:
        pristine_next = hip.DeviceArray.from_numpy(next_host)
        nxt = hip.DeviceArray(pristine_next.shape, np.int32)
    # (kept for the fused + next-hops leg of the default run, see below)
    want_next_leg = (np_dtype == np.float32 and args.engine == "perk" and not args.kslice and not args.with_next
                     and not args.no_extras and not args.no_fused_extra)
    if not want_next_leg:
        del next_host
    upd = hip.DeviceArray((engine.FWX_UPDATE_SHARDS,), np.int64).zero_()
    k_end = args.kslice if args.kslice > 0 else n
    serp = not args.no_serpentine
    relax_per_step = float(k_end) * n * n

    handle = None
    if args.engine == "fused":
        # the shipped path: a device-resident handle keeps its workspace and look-ahead stream
        handle = engine.DeviceMatrix(n, np_dtype, with_next=args.with_next, device=0)

    ev_runs = []

    def step(count=False, timed=False):
        if handle is not None:
            handle.upload_dev(pristine, pristine_next)          # D2D, blocks
            t0 = time.perf_counter()
            u = handle.solve(engine=engine.FWX_ENGINE_FUSED, k_end=k_end, count_updates=count)
            if timed:
                ev_runs.append(time.perf_counter() - t0)
            return u
        rate.copy_(pristine, stream)
        if nxt is not None:
            nxt.copy_(pristine_next, stream)
        evs = perk_solve(engine, hip, rate, nxt, n, k_end, serp, stream,
                         upd=upd if count else None, timed=timed)
        if timed:
            ev_runs.append(evs)
        return None

    updates = None
    for w in range(args.warmup):
        first = w == 0
        u = step(count=first)
        if first:
            hip.synchronize()
            updates = u if handle is not None else int(upd.numpy().sum())
    hip.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    hip.synchronize()
    dt = time.perf_counter() - t0

    out = {
        "metric": "edge-relaxations/sec (N^3/t), N=%d %s" % (n, "fp32" if es == 4 else "fp64"),
        "value": args.steps * relax_per_step / dt, "unit": "edge-relaxations/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic", "config": workload_config(args, n, k_end, serp, 1),
        "host_runtime": "HIP runtime via ctypes (no torch in the process)",
    }
    if args.kslice > 0:
        out["INVALID_debug_kslice"] = args.kslice
    u_per_launch = (updates / float(k_end)) if updates is not None else 0.0

    if handle is not None:
        passes = (k_end + 63) // 64
        kern_s = sum(ev_runs)
        # HBM side of the fused engine: one pass reads and writes the matrix once (+ next when an
        # entry changes) and reads two 64-row panels; the kernel is VALU-issue bound,
        # so this fraction says how far from the HBM roof it runs, not how good it is
        pass_bytes = 2.0 * es * n * n + 2 * 64 * es * n
        achieved = pass_bytes * passes * args.steps / kern_s / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                           "kernel": "fwx::fused_main* (64 pivots per pass)",
                           "avg_pass_us": 1e6 * kern_s / (passes * args.steps),
                           "alg_bytes_per_pass": pass_bytes, "updates_per_solve": updates,
                           "note": "VALU-issue-bound kernel: the HBM fraction is low by design "
                                   "(0.13 B of HBM traffic per relaxation against 4 B for relax_k); "
                                   "wall time of the blocking solve calls, panels included"}
        out["fused"] = {"passes_per_solve": passes,
                        "effective_GBps_at_per_k_bytes": es * relax_per_step * args.steps / kern_s / 1e9}
    else:
        launches = args.steps * k_end
        kern_ms = sum(evs[0].elapsed_time(evs[-1]) for evs, _ in ev_runs)
        evs, segs = ev_runs[-1]
        seg_us = [round(1e3 * evs[i].elapsed_time(evs[i + 1]) / max(1, segs[i + 1] - segs[i]), 1)
                  for i in range(len(evs) - 1)]
        avg_us = 1e3 * kern_ms / launches
        alg = alg_bytes_per_launch(n, es, u_per_launch, args.with_next)
        achieved = alg / (avg_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(n, args)
        cache_note = ("effective (Infinity Cache assisted): serpentine sweeps re-read the tail of the "
                      "previous launch from the 256 MiB Infinity Cache, and FETCH_SIZE counts those "
                      "hits; see serpentine_off for the plain streaming figure") if serp else \
            "plain streaming sweep (serpentine off)"
        if es * n * n <= INFINITY_CACHE_BYTES:
            cache_note = "effective: the whole matrix fits the 256 MiB Infinity Cache"
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                           "traffic_source": traffic_src, "label": cache_note,
                           "kernel": "fwx::relax_k", "avg_launch_us": avg_us,
                           "alg_bytes_per_launch": alg, "updates_per_solve": updates,
                           "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                           "avg_launch_us_by_k_sixteenth": seg_us}

    extras = not args.no_extras and not args.kslice
    timed_result = None
    if handle is None and extras and not args.no_fused_extra:
        timed_result = rate.numpy(stream)      # what the last TIMED per-k step left in HBM
    if handle is None and extras and serp:
        # the same solve with every launch sweeping in the same direction: nothing is re-read
        # from the Infinity Cache on purpose
        rate.copy_(pristine, stream)
        if nxt is not None:
            nxt.copy_(pristine_next, stream)
        evs, _ = perk_solve(engine, hip, rate, nxt, n, k_end, False, stream, timed=True)
        hip.synchronize()
        us = 1e3 * evs[0].elapsed_time(evs[-1]) / k_end
        ach = alg_bytes_per_launch(n, es, u_per_launch, args.with_next) / (us * 1e-6) / 1e9
        out["roofline"]["serpentine_off"] = {"avg_launch_us": us, "achieved": ach,
                                             "frac": ach / HBM_PEAK_GBPS, "steps": 1}
        # first-class: the un-assisted figure (every launch sweeps the same way, nothing is re-read from
        # the Infinity Cache on purpose) beside `frac`, which is the serpentine sweep's effective one
        out["roofline"]["frac_plain_stream"] = ach / HBM_PEAK_GBPS
        out["roofline"]["achieved_plain_stream"] = ach

    if handle is None and extras and es == 4 and not args.with_next and want_next_leg:
        # The per-k launch with BOTH matrices, as north_star words it ("the dense N x N rate matrix and
        # next-hop index matrix ... one launch per k"): same kernel family with next-hops carried.  One
        # solve; algorithmic bytes per launch = s*N^2 + (s + 4)*U/N + 2*s*N (the next-hop
        # matrix is only WRITTEN, 4 bytes per successful relaxation).
        pn_perk = hip.DeviceArray.from_numpy(next_host)
        nx = hip.DeviceArray(pn_perk.shape, np.int32)
        legs = {}
        for label, sflag in (("serpentine_on", True), ("serpentine_off", False)):
            rate.copy_(pristine, stream)
            nx.copy_(pn_perk, stream)
            evs, _ = perk_solve(engine, hip, rate, nx, n, k_end, sflag, stream, timed=True)
            hip.synchronize()
            us = 1e3 * evs[0].elapsed_time(evs[-1]) / k_end
            ach = alg_bytes_per_launch(n, es, u_per_launch, True) / (us * 1e-6) / 1e9
            legs[label] = {"avg_launch_us": us, "achieved": ach, "frac": ach / HBM_PEAK_GBPS}
        got_r, got_n = rate.numpy(stream), nx.numpy(stream)
        leg = {"kernel": "fwx::relax_k (rate + next)", "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "ms_per_step": legs["serpentine_on"]["avg_launch_us"] * k_end / 1e3,
               "value": relax_per_step / (legs["serpentine_on"]["avg_launch_us"] * k_end * 1e-6),
               "frac": legs["serpentine_on"]["frac"], "achieved": legs["serpentine_on"]["achieved"],
               "frac_plain_stream": legs["serpentine_off"]["frac"],
               "avg_launch_us": legs["serpentine_on"]["avg_launch_us"],
               "avg_launch_us_plain_stream": legs["serpentine_off"]["avg_launch_us"],
               "alg_bytes_per_launch": alg_bytes_per_launch(n, es, u_per_launch, True),
               "check": {"rate_digest": digest(got_r), "next_digest": digest(got_n)},
               "note": "one solve per sweep order, HIP events on the launch stream; not part of `value`"}
        gold = os.path.join(ROOT, "tests", "golden", "config4_n16384_digests.json")
        if n == 16384 and args.dist == "d1" and args.config in (0, 4) and os.path.exists(gold):
            with open(gold) as f:
                g = json.load(f)
            leg["check"]["equals_whole_oracle_solve"] = bool(leg["check"]["rate_digest"] == g["rate_digest"] and
                                                             leg["check"]["next_digest"] == g["next_digest"])
        out["per_k_with_next"] = leg
        del pn_perk, nx, got_r, got_n

    if handle is None and extras and not args.no_fused_extra:
        # Not part of `value`: the same workload on the engine fwx_solve_* / fwx_matrix_solve pick
        # by default (AUTO = fused, 64 pivots per pass, same bits), through a device-resident handle.
        h = engine.DeviceMatrix(n, np_dtype, with_next=args.with_next, device=0)
        def fused_step():
            h.upload_dev(pristine, pristine_next)
            t1 = time.perf_counter()
            h.solve()
            return time.perf_counter() - t1
        fused_step()
        ft = min(fused_step(), fused_step())
        # VALU-issue bound of the max-form kernel (rates only, f32): 8.0 cycles per pair of
        # relaxations per wave (tools/valu_rate.hip, profiles/r01_valu_issue_rates.txt), 1024 SIMDs
        valu = None
        if es == 4 and not args.with_next:
            cyc = relax_per_step / 2.0 * 8.0 / 64.0 / 1024.0
            # real-time form of the same bound: the triple (2 x v_mul_f32 + v_max3_f32) SUSTAINS
            # 1.104 ns per instruction per SIMD over a 0.3 s burst at 3 waves per SIMD, the chip holding
            # 2.33 GHz (`build/valu_rate long`, profiles/r02_valu_sustained.txt)
            stream_s = relax_per_step / 2.0 * 3 * 1.104e-9 / 64.0 / 1024.0
            valu = {"bound": "valu-issue", "cycles_per_pair_of_relaxations": 8.0,
                    "at_measured_stream_rate": {"ns_per_instruction_per_simd": 1.104, "instructions_per_pair": 3,
                                                "bound_ms_per_step": 1e3 * stream_s, "frac": stream_s / ft},
                    "at_round1_clock": {"clock_GHz": 1.92, "bound_ms_per_step": 1e3 * cyc / 1.92e9,
                                        "frac": cyc / 1.92e9 / ft},
                    "at_nominal_clock": {"clock_GHz": 2.4, "bound_ms_per_step": 1e3 * cyc / 2.4e9,
                                         "frac": cyc / 2.4e9 / ft},
                    "source": "tools/valu_rate.hip: profiles/r02_valu_sustained.txt (0.3 s bursts: 2.33 GHz "
                              "held), profiles/r02_valu_issue_rates.txt, profiles/r02_clocks.txt; whole solve "
                              "(panels and look-ahead launches included), not the main kernel alone"}
            kclk = kernel_clock_ghz("float32", False)
            if kclk:
                # the clock measured INSIDE fused_main_max on this solve shape (valu_rate's register stream holds
                # 2.33 GHz, the real kernel with its LDS operand reads less): the issue bound at that clock
                valu["at_kernel_clock"] = {"clock_GHz": kclk, "bound_ms_per_step": 1e3 * cyc / (kclk * 1e9),
                                           "frac": cyc / (kclk * 1e9) / ft,
                                           "source": "profiles/r04_shader_clock.json (tools/measure_clock.py)"}
        out["fused_engine"] = {"value": relax_per_step / ft, "unit": "edge-relaxations/s",
                               "ms_per_step": 1e3 * ft, "steps": 2, "valu_roofline": valu,
                               "note": "same workload on the default (AUTO) engine: fused (from N = 6144 a "
                                       "rates-only solve applies 128 pivots per main launch behind a two-deep "
                                       "look-ahead, else 64), bit-identical results, VALU-bound; not part of "
                                       "`value`; best of 2, blocking fwx_matrix_solve calls"}
        # cross-check of the TIMED launches: the matrix the last timed per-k step left in HBM must
        # equal the fused engine's, bit for bit (neither is the oracle; the test-suite ties both to it)
        fused_host = h.download()[0]
        it = np.uint32 if es == 4 else np.uint64
        same = bool(np.array_equal(timed_result.view(it), fused_host.view(it)))
        out["check"] = {"per_k_equals_fused_bits": same, "rate_digest": digest(fused_host),
                        "what": "matrix left in HBM by the last timed per-k step vs the fused engine's"}
        del timed_result, fused_host
        h.close()
        if want_next_leg:
            # ... and with the next-hop matrix (Algorithms.hs:55, the head of `_path`): the arg kernel.
            # Same bound as above -- one fold at the stream rate -- so the fraction says what the
            # next-hops cost on top of the rates.
            h = engine.DeviceMatrix(n, np_dtype, with_next=True, device=0)
            pn = hip.DeviceArray.from_numpy(next_host)
            def next_step():
                h.upload_dev(pristine, pn)
                t1 = time.perf_counter()
                h.solve()
                return time.perf_counter() - t1
            next_step()
            nt = min(next_step(), next_step())
            got = h.download()
            stream_s = relax_per_step / 2.0 * 3 * 1.104e-9 / 64.0 / 1024.0
            leg = {"value": relax_per_step / nt, "unit": "edge-relaxations/s", "ms_per_step": 1e3 * nt, "steps": 2,
                   "valu_roofline": {"bound": "valu-issue", "bound_ms_per_step": 1e3 * stream_s,
                                     "frac": stream_s / nt, "own_scheme": arg_kernel_own_bound(n, nt),
                                     "note": "against ONE fold at the sustained stream rate (the rates-only bound); "
                                             "the kernel's own scheme -- fold + stage tracking + compaction + "
                                             "re-scan of the moved entries -- issues ~1.45 x the fold's vector "
                                             "cycles (DESIGN.md 4.2, profiles/r03_sq_arg_main.txt)"},
                   "check": {"rate_digest": digest(got[0]), "next_digest": digest(got[1]),
                             "rates_equal_rates_only_leg": digest(got[0]) == out["check"]["rate_digest"]},
                   "note": "same workload with the next-hop matrix carried (rates + next: fused_main_arg, from N = 8192 "
                           "two passes = 128 pivots per main launch), best of 2, blocking fwx_matrix_solve calls; "
                           "not part of `value`"}
            gold = os.path.join(ROOT, "tests", "golden", "config4_n16384_digests.json")
            if n == 16384 and args.dist == "d1" and args.config in (0, 4) and os.path.exists(gold):
                with open(gold) as f:
                    g = json.load(f)
                leg["check"]["equals_whole_oracle_solve"] = bool(leg["check"]["rate_digest"] == g["rate_digest"] and
                                                                 leg["check"]["next_digest"] == g["next_digest"])
            out["fused_engine_next"] = leg
            del got, pn
            h.close()

    if want_f64 and extras:
        out["f64"] = f64_leg(engine, hip, rate64, n, stream)
    if extras and not args.kslice:
        out["reference_regime"] = reference_regime_leg(engine, not args.no_cpu_baseline)

    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rate_host, args.cpu_seconds)
    print(json.dumps(out), flush=True)


def kernel_clock_ghz(dtype_name, with_next, n=16384):
    """Shader clock measured INSIDE the fused main kernels (profiles/r04_shader_clock.json, a -DFWX_CLOCK_PROBE
    build; tools/measure_clock.py), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_shader_clock.json")) as f:
            for r in json.load(f)["runs"]:
                if r["n"] == n and r["dtype"] == dtype_name and r["next"] == with_next:
                    return float(r["shader_clock_GHz"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def arg_kernel_own_bound(n, measured_s):
    """The minimum of fused_main_arg's OWN scheme in vector-issue cycles, from the instruction mix of the
    compiled kernel (tools/isa_mix.py over `hipcc -S`, profiles/r04_arg_isa_mix.txt) priced at the issue costs
    measured on gfx950 (profiles/r02_valu_issue_rates.txt: v_mul_f32 2.15 cycles, v_max3 / v_cmp / v_cndmask and
    every other VOP3 4.1, saturated SIMD), per wave and 128 x 64 tile (8 x 4 entries per lane, 64 pivots):
      fold        32 pivot pairs x (64 v_mul_f32 + 32 v_max3_f32 + 5 loop instructions)
      tracking    4 stages x 8 rows x (4 v_cmp + 4 v_cndmask)
      compaction  32 entry slots x (5 vector + ~8 scalar instructions)
      re-scan     one batch of 64 items = 16 products, 7 v_max3, 17 v_cmp + 17 v_cndmask, address arithmetic;
                  batches per wave and tile = moved entries / 64 (11 % of 2048 entries on this solve)
      rest        staging, addressing, row stores, prologue
    against the wall time of the solve: how close the kernel runs to what its scheme can do at best (the
    `frac` beside this one is against ONE fold at the stream rate, a bound no next-hop scheme reaches)."""
    fold = 32 * (64 * 2.15 + 32 * 4.1 + 5 * 2.0)
    tracking = 4 * 8 * 8 * 4.1
    compaction = 32 * (5 * 4.1 + 8 * 1.0)
    moved = 0.11
    rescan = 260.0 * (moved * 2048 / 64.0)
    rest = 600.0
    cycles = fold + tracking + compaction + rescan + rest
    wave_tiles_per_simd = (n / 128.0) * (n / 64.0) * 4 / 1024.0
    passes = n / 64.0
    clock = 2.33e9                                     # held under VALU load (profiles/r02_valu_sustained.txt)
    bound_s = passes * wave_tiles_per_simd * cycles / clock
    out = {"cycles_per_wave_tile": {"fold": round(fold), "tracking": round(tracking), "compaction": round(compaction),
                                    "rescan_at_11pct_moved": round(rescan), "rest": round(rest), "total": round(cycles)},
           "clock_GHz": 2.33, "bound_ms_per_step": 1e3 * bound_s, "frac": bound_s / measured_s,
           "fold_share_of_own_bound": fold / cycles,
           "source": "tools/isa_mix.py, profiles/r04_arg_isa_mix.txt, profiles/r02_valu_issue_rates.txt"}
    kclk = kernel_clock_ghz("float32", True)
    if kclk:                                           # the clock measured inside the kernel itself
        at = passes * wave_tiles_per_simd * cycles / (kclk * 1e9)
        out["at_kernel_clock"] = {"clock_GHz": kclk, "bound_ms_per_step": 1e3 * at, "frac": at / measured_s,
                                  "source": "profiles/r04_shader_clock.json (tools/measure_clock.py)"}
    return out


def reference_regime_leg(engine, with_cpu):
    """The sizes the reference itself runs at (its tests stop at 4 x 4; a market of 10 exchanges x 12
    currencies has 120 vertices) through the drop-in entry point a Haskell shim would call:
    fwx_solve_f64 with next-hops and path lengths, host arrays in, host arrays out, blocking.  Median of
    100 calls (after 5), beside the CPU restatement of the reference loop on the same input (one thread,
    best of 3).  Below n ~ 32 the CPU loop wins -- the GPU call is launch + PCIe latency -- and the point
    of the GPU path there is that the reference's own cases cost a tenth of a millisecond, not to beat a
    64-entry loop; from n = 64 on the call is faster than the loop."""
    from floydwarshall_amd import synth
    rows = []
    for n in (4, 16, 64, 128, 256, 512):
        rate, nxt, hops = synth.make("d1", n, np.float64, seed=5)
        ts = []
        for _ in range(105 if n <= 128 else 35):
            r, x, h = rate.copy(), nxt.copy(), hops.copy()
            t0 = time.perf_counter()
            engine.solve(r, x, h)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[5:])
        row = {"n": n, "gpu_call_ms": round(1e3 * ts[len(ts) // 2], 4)}
        if with_cpu:
            import oracle                                   # the checker, as the CPU baseline only
            best = None
            for _ in range(3):                              # best of 3: the first call loads the checker
                er, en, eh = rate.copy(), nxt.copy(), hops.copy()
                t0 = time.perf_counter()
                oracle.relax(er, en, eh)
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            row["cpu_restatement_ms"] = round(1e3 * best, 4)
            it = np.uint64
            row["bits_equal"] = bool(np.array_equal(r.view(it), er.view(it)) and np.array_equal(x, en)
                                     and np.array_equal(h, eh))
        rows.append(row)
    return {"what": "fwx_solve_f64 (rate + next + hops; upload + solve + download, blocking), median per call; "
                    "cpu: one thread of the C restatement of Algorithms.hs:42-61 on the same input",
            "rows": rows}


def f64_leg(engine, hip, rate64, n, stream):
    """The reference's own precision (Types.hs:26, `_bestRate :: Double`), same matrix before
    rounding: per-k engine against HBM (8 B per relaxation) and the fused engine.  One solve each."""
    pristine = hip.DeviceArray.from_numpy(rate64)
    rate = hip.DeviceArray(pristine.shape, np.float64)
    rate.copy_(pristine, stream)
    perk_solve(engine, hip, rate, None, n, min(n, 512), True, stream)        # warm-up: 512 pivots
    rate.copy_(pristine, stream)
    evs, segs = perk_solve(engine, hip, rate, None, n, n, True, stream, timed=True)
    hip.synchronize()
    us = 1e3 * evs[0].elapsed_time(evs[-1]) / n
    seg_us = [round(1e3 * evs[i].elapsed_time(evs[i + 1]) / max(1, segs[i + 1] - segs[i]), 1)
              for i in range(len(evs) - 1)]
    alg = alg_bytes_per_launch(n, 8, 0.0, False)   # the s*U/N term (~0.2 %) is left out: conservative
    ach = alg / (us * 1e-6) / 1e9
    res = {"per_k": {"ms_per_step": us * n / 1e3, "value": float(n) ** 3 / (us * n * 1e-6),
                     "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBPS, "kernel": "fwx::relax_k<double>",
                                  "avg_launch_us": us, "alg_bytes_per_launch": alg,
                                  "avg_launch_us_by_k_sixteenth": seg_us,
                                  "label": "effective (Infinity Cache assisted, serpentine on)"}}}
    h = engine.DeviceMatrix(n, np.float64, with_next=False, device=0)
    times = []
    for _ in range(2):
        h.upload_dev(pristine)
        t1 = time.perf_counter()
        h.solve()
        times.append(time.perf_counter() - t1)
    ft = min(times)
    fused_host = h.download()[0]
    same = bool(np.array_equal(fused_host.view(np.uint64), rate.numpy(stream).view(np.uint64)))
    f64_digest = digest(fused_host)
    del fused_host
    # f64 max form: v_mul_f64 + v_max_f64; the pair SUSTAINS 1.907 ns per instruction per SIMD over a
    # 0.2 s burst at 2-3 waves per SIMD, 2.38 GHz held (profiles/r02_valu_sustained.txt)
    cyc = float(n) ** 3 * 8.7 / 64.0 / 1024.0
    stream_s = float(n) ** 3 * 2 * 1.907e-9 / 64.0 / 1024.0
    res["fused"] = {"ms_per_step": 1e3 * ft, "value": float(n) ** 3 / ft,
                    "valu_roofline": {"cycles_per_relaxation": 8.7,
                                      "at_measured_stream_rate": {"ns_per_instruction_per_simd": 1.907,
                                                                  "bound_ms_per_step": 1e3 * stream_s,
                                                                  "frac": stream_s / ft},
                                      "at_round1_clock": {"clock_GHz": 1.92, "frac": cyc / 1.92e9 / ft},
                                      "at_nominal_clock": {"clock_GHz": 2.4, "frac": cyc / 2.4e9 / ft}},
                    "per_k_equals_fused_bits": same}
    kclk = kernel_clock_ghz("float64", False)
    if kclk:                                           # the clock measured inside fused_main_max_f64 (profiles/r04_shader_clock.json)
        res["fused"]["valu_roofline"]["at_kernel_clock"] = {"clock_GHz": kclk, "bound_ms_per_step": 1e3 * cyc / (kclk * 1e9),
                                                           "frac": cyc / (kclk * 1e9) / ft}
    h.close()
    res["check"] = {"rate_digest": f64_digest}
    gold = os.path.join(ROOT, "tests", "golden", "config4_n16384_f64_digests.json")
    if n == 16384 and os.path.exists(gold):
        with open(gold) as f:
            g = json.load(f)
        # (the committed digest is of ONE WHOLE f64 oracle solve of this matrix: 265 s on 16 host threads)
        res["check"]["equals_whole_oracle_solve"] = bool(f64_digest == g["rate_digest"])
    res["note"] = "not part of `value`; same D1 matrix before rounding to f32; 1 solve per engine"
    return res


# ------------------------------------------------------------------------------------------------
# N > 1, ONE process: the row-partitioned handle behind the C ABI (torch-free)
# ------------------------------------------------------------------------------------------------
def run_multi(args):
    """`python bench.py --gpus N` without a launcher: the call a reference host binds
    (ProcessRequests.hs:82-84 -> floydWarshall, Algorithms.hs:19-20) with a device list --
    fwx_matrix_create_multi over devices 0..N-1, FWX_XCHG_AUTO (= RCCL between distinct devices), one
    host thread.  Same workload, steps and restore-from-pristine contract as N = 1: the handle keeps
    the uploaded input on the devices and every step restores it (device-to-device, inside the timed
    region) before the solve."""
    from floydwarshall_amd import engine, hip
    assert "torch" not in sys.modules, "the single-process benchmark must stay torch-free"
    devs = args.device_list
    world = len(devs)
    have = hip.device_count()
    if have < 1:
        raise SystemExit("bench.py: no HIP device (libfwx has no CPU fallback)")
    if max(devs) >= have:
        raise SystemExit("bench.py: --gpus %d / --devices %s needs device ordinals < %d (this machine has "
                         "%d HIP device%s)" % (world, ",".join(map(str, devs)), have, have,
                                               "" if have == 1 else "s"))
    logical = len(set(devs)) != world
    n = args.n
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    es = np.dtype(np_dtype).itemsize
    rate64, next_host = make_input(args, n)
    rate_host = rate64 if np_dtype == np.float64 else rate64.astype(np.float32)
    del rate64
    if args.kslice:
        raise SystemExit("--kslice is a single-GPU debug option")
    serp = not args.no_serpentine
    relax_per_step = float(n) ** 3
    base_line = {"metric": "edge-relaxations/sec (N^3/t), N=%d %s" % (n, "fp32" if es == 4 else "fp64"),
                 "unit": "edge-relaxations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                 "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
                 "data": "synthetic", "config": workload_config(args, n, n, serp, world, "row-block x%d over HIP "
                                                                "devices %s, ONE process" % (world, devs))}
    dog = Watchdog(args.step_timeout, base_line)
    dog.arm("create the partitioned handle (%s exchange) + upload" % args.exchange)
    h, xinfo = create_multi_handle(engine, n, np_dtype, args.with_next, devs, args.exchange)
    h.keep_input()
    h.upload(rate_host, next_host if args.with_next else None)
    dog.disarm()
    del next_host
    eng = engine.FWX_ENGINE_PERK if args.engine == "perk" else engine.FWX_ENGINE_FUSED

    def step(count=False, which=None):
        h.patch_input([], np.empty(0, dtype=np_dtype))      # restore the kept input on every partition
        return h.solve(engine=eng if which is None else which, serpentine=serp, count_updates=count)

    def guarded_first_step():
        """The first solve is where a communicator that initialised but cannot carry a panel shows:
        FWX_ERR_RCCL here, under `--exchange auto`, rebuilds the handle over peer copies once."""
        nonlocal h
        try:
            return step(count=True)
        except engine.FwxError as err:
            if args.exchange != "auto" or err.status != engine.FWX_ERR_RCCL or xinfo["rccl_error"]:
                raise
            xinfo["rccl_error"] = {"status": int(err.status), "message": str(err), "where": "first solve"}
            h.close()
            h = engine.DeviceMatrix(n, np_dtype, with_next=args.with_next, devices=devs, exchange=engine.FWX_XCHG_PEER)
            h.keep_input()
            h.upload(rate_host, make_input(args, n)[1] if args.with_next else None)
            return step(count=True)

    updates = None
    dog.arm("first solve (counted)")
    updates = guarded_first_step()          # always one counted solve: U and the transport check
    dog.disarm()
    for w in range(1, args.warmup):
        dog.arm("warm-up step %d" % w)
        step()
        dog.disarm()
    parts, transport = h.parts()
    ranks = h.comm_ranks()
    hip.synchronize(devs)
    t0 = time.perf_counter()
    for i in range(args.steps):
        dog.arm("timed step %d of %d" % (i + 1, args.steps))
        step()
        dog.disarm()
    hip.synchronize(devs)
    dt = time.perf_counter() - t0

    bounds = [n * p // world for p in range(world + 1)]
    slab_bytes = es * n * max(bounds[p + 1] - bounds[p] for p in range(world))
    partition = ("row-block x%d over HIP devices %s, ONE process (fwx_matrix_create_multi), 64-pivot "
                 "snapshot panels on %s" % (world, devs, "RCCL" if transport == engine.FWX_XCHG_RCCL
                                            else "peer / device-to-device copies"))
    out = {
        "metric": "edge-relaxations/sec (N^3/t), N=%d %s" % (n, "fp32" if es == 4 else "fp64"),
        "value": args.steps * relax_per_step / dt, "unit": "edge-relaxations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic", "config": workload_config(args, n, n, serp, world, partition),
        "host_runtime": "HIP runtime via ctypes (no torch in the process)",
        "exchange": {"transport": "rccl" if transport == engine.FWX_XCHG_RCCL else "peer",
                     "requested": xinfo["requested"], "rccl_error": xinfo["rccl_error"],
                     "rccl_ranks_in_communicator": ranks, "partitions": parts,
                     "distinct_devices": len(set(devs))},
    }
    dog.set_valid_line(out)         # `value` stands: whatever hangs from here on cannot cost the line
    if args.warmup < 1:
        out["warmup_note"] = "one counted solve ran before the timed region regardless of --warmup 0"

    def timing_leg(which, label):
        """One more solve with per-step event timings switched on (two event records per span: kept out
        of the timed region): is a step bound by the slab sweep or by the panel chain?"""
        dog.arm("event-timed solve (%s)" % label)
        h.set_timing(True)
        step(which=which)
        t = h.timing()
        h.set_timing(False)
        dog.disarm()
        return {"avg_bulk_us": t["bulk_us"], "avg_bulk_mean_us": t["bulk_mean_us"],
                "avg_lookahead_us": t["lookahead_us"], "avg_panel_us": t["panel_us"],
                "avg_exchange_us": t["exchange_us"], "avg_chain_us": t["chain_us"],
                "chain_over_bulk": t["chain_over_bulk"], "steps": t["steps"],
                "pivots_per_step": t["pivots_per_step"],
                "avg_bulk_us_per_block": t["bulk_us"] * 64.0 / max(1, t["pivots_per_step"])}

    if not args.no_extras:
        try:
            out["exchange"].update(timing_leg(None, args.engine))
            out["exchange"]["timing_note"] = (
                "HIP events on the streams the work runs on, one extra untimed solve: bulk = the slab sweep of a "
                "step (mean over steps of the max over partitions), chain = what the next step waits for besides it "
                "(look-ahead rows + owner's panel kernel + exchange; whole side chain under the pair schedule); "
                "chain_over_bulk > 1 = bound by the panel chain (DESIGN.md section 5)")
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    if not args.no_extras and not args.no_fused_extra and args.engine == "perk":
        try:
            # Not part of `value`: the same workload on the engine AUTO picks (fused: 128 pivots per main launch
            # behind a two-deep look-ahead where the partitions allow), same handle, best of 2
            def fused_step():
                h.patch_input([], np.empty(0, dtype=np_dtype))
                t1 = time.perf_counter()
                h.solve(engine=engine.FWX_ENGINE_FUSED)
                return time.perf_counter() - t1
            dog.arm("fused-engine leg")
            fused_step()
            ft = min(fused_step(), fused_step())
            dog.disarm()
            leg = {"value": relax_per_step / ft, "unit": "edge-relaxations/s", "ms_per_step": 1e3 * ft, "steps": 2,
                   "note": "same workload, same handle, fwx_opts.engine = FUSED (what AUTO runs); solve only (the "
                           "restore of the kept input is outside this clock); not part of `value`"}
            got = h.download()
            leg["check"] = {"rate_digest": digest(got[0])}
            del got
            leg["exchange"] = timing_leg(engine.FWX_ENGINE_FUSED, "fused")
            out["fused_engine"] = leg
            step()                               # the line's `check` below looks at the per-k result again
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    if logical:
        out["INVALID_logical_partitions"] = ("%d partitions time-share %d device(s): a rehearsal of the "
                                             "N > 1 code path, not a scaling number" % (world, len(set(devs))))
    # Aggregate HBM figure: every partition streams its slab once per pivot (per-k engine) --
    # algorithmic bytes of the whole solve over the wall time of the step (restore, panels, exchange
    # and look-ahead included) against world x 8 TB/s (x distinct devices for a rehearsal).  A slab that
    # fits the Infinity Cache is served from it: "effective".
    peak = HBM_PEAK_GBPS * len(set(devs))
    if args.engine == "perk":
        u_solve = float(updates or 0)
        alg_total = es * relax_per_step + (es + (4 if args.with_next else 0)) * u_solve + 2.0 * es * n * n
        achieved = alg_total * args.steps / dt / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                           "frac": achieved / peak, "traffic": None,
                           "kernel": "fwx::relax_k on %d row-block slabs" % world,
                           "alg_bytes_per_solve": alg_total, "updates_per_solve": updates,
                           "label": ("effective (the %d MiB slab of each partition fits the 256 MiB Infinity "
                                     "Cache: not an HBM-traffic claim)" % (slab_bytes >> 20))
                           if slab_bytes <= INFINITY_CACHE_BYTES else "aggregate over partitions",
                           "note": "algorithmic bytes of the solve (s*N^3 + s*U + 2*s*N^2) / wall time per "
                                   "step, all partitions; includes the device-to-device restore of the "
                                   "input, the panel kernels and the exchange"}
    else:
        pass_bytes = 2.0 * es * n * n + 2 * 64 * es * n * world
        achieved = pass_bytes * ((n + 63) // 64) * args.steps / dt / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                           "frac": achieved / peak, "traffic": None,
                           "kernel": "fwx::fused_main* on %d row-block slabs" % world,
                           "updates_per_solve": updates,
                           "note": "VALU-issue-bound kernel: low HBM fraction by design"}
    if not args.no_extras:
        try:
            # tie the timed result to the committed whole-oracle digests where they exist (N = 16384 f32 D1)
            got = h.download()
            out["check"] = {"rate_digest": digest(got[0])}
            if args.with_next:
                out["check"]["next_digest"] = digest(got[1])
            gold = os.path.join(ROOT, "tests", "golden", "config4_n16384_digests.json")
            if n == 16384 and es == 4 and args.dist == "d1" and args.config in (0, 4) and os.path.exists(gold):
                with open(gold) as f:
                    g = json.load(f)
                out["check"]["equals_whole_oracle_solve"] = bool(
                    out["check"]["rate_digest"] == g["rate_digest"] and
                    (not args.with_next or out["check"]["next_digest"] == g["next_digest"]) and
                    (updates is None or updates == g["U"]))
            del got
            if "fused_engine" in out:
                out["fused_engine"]["check"]["equals_timed_engine_bits"] = bool(
                    out["fused_engine"]["check"]["rate_digest"] == out["check"]["rate_digest"])
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    dog.arm("destroy the handle")
    h.close()
    dog.disarm()
    dog.stop()
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rate_host, min(args.cpu_seconds, 8.0))
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------
# N > 1: one process per GPU, torch.distributed over RCCL
# ------------------------------------------------------------------------------------------------
def run_dist(args, world, rank, local_rank):
    """One process per GPU under torch.distributed.run.  Driver "part" (default): every rank holds ONE partition
    of libfwx's own partitioned handle (fwx_matrix_create_part) -- the schedule, kernels and event timings are
    the library's, exactly the code the one-process form runs, and the panel broadcast is handed back to
    torch.distributed from libfwx's exchange callback (floydwarshall_amd.dist.PartMatrix).  Driver "legacy":
    the older Python schedule over the device-pointer entry points (dist.solve_partitioned); with the "part"
    driver it still runs once, untimed, as a cross-check and for comparison (`legacy_driver`)."""
    import datetime
    import torch
    import torch.distributed as dist
    from floydwarshall_amd import dist as fwdist

    dev_index = local_rank % torch.cuda.device_count()   # one rank per GPU on a real node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a rank that dies must fail the job quickly, not leave the others waiting in a collective
    tmo = datetime.timedelta(seconds=240)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev, timeout=tmo)
    else:
        dist.init_process_group("gloo", timeout=tmo)

    n = args.n
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    es = np.dtype(np_dtype).itemsize
    k_end = n
    relax_per_step = float(k_end) * n * n
    ENG = {"perk": fwdist.engine.FWX_ENGINE_PERK, "fused": fwdist.engine.FWX_ENGINE_FUSED}
    # every rank runs a watchdog (a hung collective hangs all of them); rank 0 writes the line
    dog = Watchdog(args.step_timeout,
                   {"metric": "edge-relaxations/sec (N^3/t), N=%d %s" % (n, "fp32" if es == 4 else "fp64"),
                    "unit": "edge-relaxations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
                    "data": "synthetic", "config": workload_config(args, n, k_end, True, world)},
                   emit=rank == 0)

    driver, driver_note, ph = args.driver, None, None
    if driver == "part":
        try:        # (fails the same way on every rank, before any collective: an old library, no device)
            ph = fwdist.PartMatrix(n, np_dtype, rank, world, with_next=args.with_next, device=dev_index)
        except Exception as err:          # noqa: BLE001 -- whatever it is, the legacy driver can still run
            driver, driver_note = "legacy", "PartMatrix could not be created (%s): legacy driver" % err
    legacy_bounds = fwdist.row_bounds(n, world)
    bounds = ph.bounds() if ph is not None else legacy_bounds
    r0, r1 = bounds[rank], bounds[rank + 1]
    same_cut = bounds == legacy_bounds      # (the library cuts on multiples of 64: N = 16384 at 2 / 4 / 8 ranks agree)
    rate64, next_host = make_input(args, n)
    pristine = torch.from_numpy(rate64[r0:r1].astype(np_dtype)).to(dev)
    pristine_next = torch.from_numpy(next_host[r0:r1]).to(dev) if args.with_next else None
    del rate64, next_host
    rate = torch.empty_like(pristine)                       # the legacy driver solves in place, in torch tensors
    nxt = torch.empty_like(pristine_next) if args.with_next else None
    backend = fwdist.HipBackend(args.engine)

    def legacy_step(bk=None, timer=None):
        rate.copy_(pristine)
        if nxt is not None:
            nxt.copy_(pristine_next)
        fwdist.solve_partitioned(rate, n, rank, world, nxt=nxt, block=args.block, backend=bk or backend,
                                 timer=timer)

    def step(engine_name=None):
        """Restore the pristine slab (device to device) + one full solve, both inside whatever clock runs."""
        if ph is not None:
            ph.upload_dev(pristine, pristine_next)
            ph.solve(engine=ENG[engine_name or args.engine])
        else:
            legacy_step(bk=None if engine_name in (None, args.engine) else fwdist.HipBackend(engine_name))

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        dog.arm("warm-up step %d" % (w + 1))
        step()
        torch.cuda.synchronize()
        dog.disarm()
    dog.arm("barrier before the timed region")
    fence()
    dog.disarm()
    t0 = time.perf_counter()
    dog.arm("the %d timed steps" % args.steps)
    for _ in range(args.steps):
        step()
    fence()
    dog.disarm()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    def max_over_ranks(values):
        v = torch.tensor(values, dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return [float(x) for x in v.tolist()]

    def timing_leg(engine_name):
        """One more, untimed solve with per-step event spans on every rank: the slab sweep of a step against
        what the next step waits for besides it, each as the MAX over ranks of the rank's figure."""
        dog.arm("event-timed solve (%s)" % engine_name)
        if ph is not None:
            ph.set_timing(True)
            step(engine_name)
            tm = ph.timing()
            ph.set_timing(False)
            bulk, chain, xch, pan, la = max_over_ranks([tm["bulk_us"], tm["chain_us"], tm["exchange_us"],
                                                        tm["panel_us"], tm["lookahead_us"]])
            pps, steps = tm["pivots_per_step"], tm["steps"]
        else:
            tm = fwdist.StepTimer(on_gpu=True)
            legacy_step(bk=None if engine_name == args.engine else fwdist.HipBackend(engine_name), timer=tm)
            sm = tm.summary()
            bulk, la, pan, xch = max_over_ranks([sm[k][0] for k in fwdist.StepTimer.KINDS])
            chain, pps, steps = la + pan + xch, args.block, sm["bulk"][1]
        dog.disarm()
        return {"avg_bulk_us": bulk, "avg_lookahead_us": la, "avg_panel_us": pan, "avg_exchange_us": xch,
                "avg_chain_us": chain, "chain_over_bulk": chain / bulk if bulk > 0 else None,
                "pivots_per_step": pps, "steps": steps, "avg_bulk_us_per_block": bulk * 64.0 / max(1, pps)}

    out = {
        "metric": "edge-relaxations/sec (N^3/t), N=%d %s" % (n, "fp32" if es == 4 else "fp64"),
        "value": args.steps * relax_per_step / dt, "unit": "edge-relaxations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic", "config": workload_config(args, n, k_end, True, world),
        "driver": {"name": driver, "note": driver_note,
                   "what": "part: one partition per process behind the C ABI (fwx_matrix_create_part), libfwx's own "
                           "schedule, panels broadcast by torch.distributed from its exchange callback; legacy: "
                           "dist.solve_partitioned, the Python schedule over the device-pointer entry points"},
    }
    if args.backend != "nccl":
        out["INVALID_rehearsal_backend"] = args.backend
    # Aggregate HBM figure: every rank streams its slab once per pivot (per-k engine) -- algorithmic
    # bytes of the whole solve over the wall time of the step (exchange and look-ahead included),
    # against world x 8 TB/s.  A slab that fits the Infinity Cache is served from it: "effective".
    slab_bytes = es * n * max(bounds[p + 1] - bounds[p] for p in range(world))
    if args.engine == "perk":
        alg_total = es * relax_per_step + 2.0 * es * n * n
        achieved = alg_total * args.steps / dt / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS * world,
                           "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBPS * world), "traffic": None,
                           "kernel": "fwx::relax_k on %d row-block slabs" % world,
                           "label": ("effective (the %d MiB slab of each rank fits the 256 MiB Infinity "
                                     "Cache: not an HBM-traffic claim)" % (slab_bytes >> 20))
                           if slab_bytes <= INFINITY_CACHE_BYTES else "aggregate over ranks",
                           "note": "algorithmic bytes of the solve (s*N^3 + 2*s*N^2; the s*U term, "
                                   "~0.2 %, is left out) / wall time per step, all ranks; includes the "
                                   "device-to-device restore and the panel exchange"}
    else:
        pass_bytes = 2.0 * es * n * n + 2 * 64 * es * n * world
        achieved = pass_bytes * ((k_end + 63) // 64) * args.steps / dt / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS * world,
                           "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBPS * world), "traffic": None,
                           "kernel": "fwx::fused_main* on %d row-block slabs" % world,
                           "note": "VALU-issue-bound kernel: low HBM fraction by design"}
    out["exchange"] = {"transport": "rccl (torch.distributed)" if args.backend == "nccl" else args.backend,
                       "ranks_in_process_group": dist.get_world_size()}
    dog.set_valid_line(out)         # `value` stands: whatever hangs from here on cannot cost the line
    if not args.no_extras:
        try:
            out["exchange"].update(timing_leg(args.engine))
            out["exchange"]["timing_note"] = (
                "events on the streams the work runs on, one extra untimed solve: bulk = the slab sweep of a step, "
                "chain = what the next step waits for besides it (look-ahead rows + owner's panel + broadcast; the "
                "whole side chain under the pair schedule); chain_over_bulk > 1 = bound by the panel chain "
                "(DESIGN.md section 5)")
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    if not args.no_extras and not args.no_fused_extra and args.engine == "perk":
        try:
            # Not part of `value`: the same workload on the engine AUTO picks (fused; the 128-pivot pair schedule
            # under the "part" driver), same handle, solve only, best of 2, max over ranks
            dog.arm("fused-engine leg")
            times = []
            for i in range(3):
                if ph is not None:
                    ph.upload_dev(pristine, pristine_next)
                fence()
                t1 = time.perf_counter()
                if ph is not None:
                    ph.solve(engine=ENG["fused"])
                else:
                    legacy_step(bk=fwdist.HipBackend("fused"))
                fence()
                times.append(time.perf_counter() - t1)
            ft = min(max_over_ranks(times[1:]))
            dog.disarm()
            out["fused_engine"] = {"value": relax_per_step / ft, "unit": "edge-relaxations/s", "ms_per_step": 1e3 * ft,
                                   "steps": 2, "exchange": timing_leg("fused"),
                                   "note": "same workload, engine = FUSED (what AUTO runs), same driver; %s; max over "
                                           "ranks, best of 2; not part of `value`"
                                           % ("solve only" if ph is not None else "restore included")}
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    if not args.no_extras:
        try:
            # What the ranks computed, checked on rank 0: every rank's slab of the last timed step's result gathered
            # (device tensors over RCCL; host tensors under the gloo rehearsal), digested in row order, and -- for the
            # headline matrix -- compared with the digest of the whole CPU-oracle solve (tests/golden/)
            dog.arm("result check (gather of the slabs on rank 0)")
            step()                                   # the timed engine's result again
            torch.cuda.synchronize()
            if ph is not None:
                got = ph.download()
                mine, mine_next = torch.from_numpy(got[0]), (torch.from_numpy(got[1]) if args.with_next else None)
                del got
            else:
                mine, mine_next = rate.cpu(), (nxt.cpu() if nxt is not None else None)
            rows_max = max(bounds[p + 1] - bounds[p] for p in range(world))

            def gather_rows(t):
                buf = torch.zeros((rows_max, n), dtype=t.dtype)
                buf[:r1 - r0] = t
                if args.backend == "nccl":
                    buf = buf.to(dev)
                parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
                dist.gather(buf, parts, dst=0)
                if rank != 0:
                    return None
                return [parts[p][:bounds[p + 1] - bounds[p]].cpu().numpy() for p in range(world)]

            slabs = gather_rows(mine)
            slabs_next = gather_rows(mine_next) if mine_next is not None else None
            if rank == 0:
                chk = {"rate_digest": digest(slabs), "what": "the ranks' slabs of the timed engine's result, gathered on "
                                                               "rank 0 and digested in row order"}
                if slabs_next is not None:
                    chk["next_digest"] = digest(slabs_next)
                gold = os.path.join(ROOT, "tests", "golden", "config4_n16384_digests.json")
                if n == 16384 and es == 4 and args.dist == "d1" and args.config in (0, 4) and os.path.exists(gold):
                    with open(gold) as f:
                        g = json.load(f)
                    chk["equals_whole_oracle_solve"] = bool(chk["rate_digest"] == g["rate_digest"] and
                                                            (slabs_next is None or chk["next_digest"] == g["next_digest"]))
                out["check"] = chk
            del slabs, slabs_next, mine, mine_next
            dog.disarm()
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    if not args.no_extras and ph is not None and same_cut:
        try:
            # the older Python driver once, untimed region of its own: a cross-check of the two schedules (bit for
            # bit, every rank's slab) and their per-k times side by side
            dog.arm("legacy driver leg (dist.solve_partitioned)")
            step()                                   # the timed engine's result again, in the handle
            mine = torch.from_numpy(ph.download()[0]).to(dev)
            lt = []
            for i in range(2):
                fence()
                t1 = time.perf_counter()
                legacy_step()
                fence()
                lt.append(time.perf_counter() - t1)
            same = torch.tensor([1 if torch.equal(mine, rate) else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            ls = max_over_ranks([lt[1]])[0]
            dog.disarm()
            out["legacy_driver"] = {"value": relax_per_step / ls, "ms_per_step": 1e3 * ls, "steps": 1,
                                    "equals_timed_driver_bits": bool(int(same.item())),
                                    "note": "dist.solve_partitioned, same engine, restore included; not part of `value`"}
            del mine
        except Exception as err:      # noqa: BLE001 -- an optional leg must not cost the line
            out.setdefault("extras_failed", []).append({"leg": dog._label, "error": repr(err)[:400]})
            dog.disarm()
    if ph is not None:
        ph.close()
    dog.disarm()
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(make_input(args, n)[0].astype(np_dtype),
                                               min(args.cpu_seconds, 8.0))
        print(json.dumps(out), flush=True)
    # the line is out: a rank that is gone (its own watchdog ended it during an optional leg) must not leave the
    # others in this barrier for ever
    dog.line_printed()
    dog.arm("final barrier")
    dist.barrier()
    dist.destroy_process_group()
    dog.stop()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:                           # launched by torch.distributed.run: one process per GPU
        args.gpus = world
        run_dist(args, world, rank, local_rank)
    elif len(args.device_list) > 1 or args.devices:
        run_multi(args)                     # no launcher: ONE process through the partitioned C-ABI handle
    else:
        run_single(args)


if __name__ == "__main__":
    main()
