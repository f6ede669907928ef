#!/usr/bin/env python3
"""bench.py -- edge-relaxations/s (N^3/t) of the max-product Floyd-Warshall hot path on MI355X.

Contract (one JSON line from rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE FULL SOLVE (all N pivots of runAlgo, /root/reference/src/lib/Algorithms.hs:42-61)
of the synthetic dense matrix, restarted from the pristine input (a device-to-device copy inside
the timed region, ~0.03 % of a step).  Workload = BASELINE.json's metric configuration:
N = 16384, fp32, dense D1 input (floydwarshall_amd/synth.py), matrices resident in HBM.

  value     = K * N^3 / t      t = wall time of the K steps, barrier + synchronize on both sides,
                               max over ranks
  roofline  = per-k kernel `relax_k` against HBM: algorithmic bytes per launch
              (s*N^2 + s*U/N + 2*s*N) / average launch duration measured
              live with HIP events on the launch stream over the timed region
  cpu_baseline = the oracle's multithreaded dense loop (a C restatement of the reference loop --
              the Haskell reference cannot be built here) on a bounded k-slice of the same matrix
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC; RCCL needs it before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SEGMENTS = 16
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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
    ap.add_argument("--block", type=int, default=64, help="pivots per broadcast (N > 1)")
    ap.add_argument("--no-serpentine", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-extra", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (gloo only to rehearse the N > 1 code path "
                         "with several ranks on one GPU; never a performance number)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
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
    return args


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
    """HBM bytes per relax_k launch from the committed rocprofv3 PMC passes (profiles/), which
    cannot be collected inside this process.  Only quoted for the exact configuration they were
    measured on (N=16384 fp32 rates-only single GPU); otherwise null."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if n != 16384 or args.dtype != "f32" or args.with_next or args.kslice or not os.path.exists(path):
        return None, None
    with open(path) as f:
        t = json.load(f)
    return t["traffic_bytes_per_launch"], ("profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE "
                                           "and --pmc WRITE_SIZE passes, FETCH_SIZE x2 (gfx950), KiB units")


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from floydwarshall_amd import engine, synth
    from floydwarshall_amd import dist as fwdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    dev_index = local_rank % torch.cuda.device_count()   # one rank per GPU on a real node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a rank that dies must fail the job quickly, not leave the others waiting in a collective
        tmo = datetime.timedelta(seconds=240)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    n = args.n
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    es = np.dtype(np_dtype).itemsize
    cfg_index = {0: 3, 2: 1, 3: 2, 4: 3, 5: 4}[args.config]  # seed = BASE_SEED + configs[] index
    rate_host, next_host = synth.GENERATORS[args.dist](n, np_dtype, synth.BASE_SEED + cfg_index)

    bounds = fwdist.row_bounds(n, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    pristine = torch.from_numpy(rate_host[r0:r1]).to(dev)
    rate = torch.empty_like(pristine)
    pristine_next = nxt = None
    if args.with_next:
        pristine_next = torch.from_numpy(next_host[r0:r1]).to(dev)
        nxt = torch.empty_like(pristine_next)
    del next_host
    if not (rank == 0 and world == 1 and not args.no_cpu_baseline):
        rate_host = None
    upd = torch.zeros(engine.FWX_UPDATE_SHARDS, dtype=torch.int64, device=dev)
    k_end = args.kslice if args.kslice > 0 else n
    serp = not args.no_serpentine

    dist_backend = fwdist.HipBackend(args.engine) if world > 1 else None
    ev_pairs = []
    fused_ws = None
    if args.engine == "fused" and world == 1:
        fused_ws = engine.FusedWorkspace(n, n, rate.dtype, dev, with_next=args.with_next)

    def step(count=False, timed=False):
        rate.copy_(pristine)
        if nxt is not None:
            nxt.copy_(pristine_next)
        if world == 1 and fused_ws is not None:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if count:
                engine.dev_solve_fused(rate, n, 0, k_end, next_t=nxt, ws=fused_ws, updates_t=upd)
            else:
                engine.dev_solve(rate, next_t=nxt, engine=engine.FWX_ENGINE_FUSED, k_end=k_end)
            if timed:
                e1.record()
                ev_pairs.append(([e0, e1], [0, k_end]))
        elif world == 1:
            # the pivots are issued in SEGMENTS back-to-back launches with a HIP event between
            # segments (no synchronisation): per-segment launch time shows how the cost moves with k
            segs = [k_end * i // SEGMENTS for i in range(SEGMENTS + 1)] if timed else [0, k_end]
            evs = []
            for a, b in zip(segs[:-1], segs[1:]):
                if timed:
                    evs.append(torch.cuda.Event(enable_timing=True))
                    evs[-1].record()
                if b > a:
                    engine.dev_relax(rate, n, 0, a, b, next_t=nxt, serpentine=serp,
                                     updates_t=upd if count else None)
            if timed:
                evs.append(torch.cuda.Event(enable_timing=True))
                evs[-1].record()
                ev_pairs.append((evs, segs))
        else:
            fwdist.solve_partitioned(rate, n, rank, world, nxt=nxt, block=args.block,
                                     backend=dist_backend)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    updates = None
    for w in range(args.warmup):
        first = (w == 0 and world == 1)
        if first:
            upd.zero_()
        step(count=first)
        if first:
            torch.cuda.synchronize()
            updates = int(upd.sum().item())
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    relax_per_step = float(k_end) * n * n
    value = args.steps * relax_per_step / dt
    out = {
        "metric": "edge-relaxations/sec (N^3/t), N=%d %s" % (n, "fp32" if es == 4 else "fp64"),
        "value": value, "unit": "edge-relaxations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "N=%d dense %s rate matrix (%s), full solve = %d pivot steps per "
                               "step, %s%s" % (n, args.dtype, args.dist.upper(), k_end,
                                               "per-k engine" if args.engine == "perk"
                                               else "fused engine (64 pivots per pass)",
                                               ", with next-hop matrix" if args.with_next else ""),
                   "n": n, "engine": "perk", "serpentine": serp,
                   "partition": "single GPU" if world == 1 else
                   "row-block x%d, %d-pivot snapshot panels broadcast on RCCL" % (world, args.block)},
    }
    if args.kslice > 0:
        out["INVALID_debug_kslice"] = args.kslice
    if world > 1 and args.backend != "nccl":
        out["INVALID_rehearsal_backend"] = args.backend

    out["config"]["engine"] = args.engine
    if world == 1 and ev_pairs and args.engine == "fused":
        passes = args.steps * ((k_end + 63) // 64)
        kern_ms = sum(evs[0].elapsed_time(evs[-1]) for evs, _ in ev_pairs)
        out["fused"] = {"passes_per_solve": (k_end + 63) // 64, "avg_pass_us": 1e3 * kern_ms / passes,
                        "effective_GBps_at_4B_per_relaxation": 4.0 * relax_per_step * args.steps / (kern_ms * 1e-3) / 1e9,
                        "note": "VALU-issue-bound kernel (8.0 cycles per pair of relaxations, DESIGN 4.2); the GB/s figure is "
                                "'effective' (algorithmic bytes of the per-k form), not an HBM roofline "
                                "fraction", "updates_per_solve": updates}
    elif world == 1 and ev_pairs:
        launches = args.steps * k_end
        kern_ms = sum(evs[0].elapsed_time(evs[-1]) for evs, _ in ev_pairs)
        evs, segs = ev_pairs[-1]
        seg_us = [round(1e3 * evs[i].elapsed_time(evs[i + 1]) / max(1, segs[i + 1] - segs[i]), 1)
                  for i in range(len(evs) - 1)]
        avg_us = 1e3 * kern_ms / launches
        u_per_launch = (updates / float(k_end)) if updates is not None else 0.0
        # B_alg = s*N^3 + s*U + 2*s*N^2 per solve (+4*U with next)
        alg_bytes = es * n * n + (es + (4 if args.with_next else 0)) * u_per_launch + 2 * es * n
        achieved = alg_bytes / (avg_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(n, args)
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                           "traffic_source": traffic_src,
                           "kernel": "fwx::relax_k", "avg_launch_us": avg_us,
                           "alg_bytes_per_launch": alg_bytes, "updates_per_solve": updates,
                           "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                           "avg_launch_us_by_k_sixteenth": seg_us}
    if world == 1 and args.engine == "perk" and not args.kslice and not args.no_fused_extra:
        # Not part of `value`: the same workload on the fused engine (64 pivots per pass, same
        # bits), measured after the timed region.  See DESIGN.md section 4.2.
        ws = engine.FusedWorkspace(n, n, rate.dtype, dev, with_next=args.with_next)
        def fused_step():
            rate.copy_(pristine)
            if nxt is not None:
                nxt.copy_(pristine_next)
            engine.dev_solve(rate, next_t=nxt, engine=engine.FWX_ENGINE_FUSED)
        fused_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            fused_step()
        torch.cuda.synchronize()
        ft = (time.perf_counter() - t1) / 2
        # VALU-issue bound of the max-form kernel (rates only, f32): 8.0 cycles per pair of
        # relaxations per wave (tools/valu_rate.hip, profiles/r01_valu_issue_rates.txt), 1024 SIMDs,
        # at the 1.92 GHz the chip holds under this load
        valu = None
        if es == 4 and not args.with_next:
            bound_s = relax_per_step / 2.0 * 8.0 / 64.0 / 1024.0 / 1.92e9
            valu = {"bound": "valu-issue", "cycles_per_pair_of_relaxations": 8.0, "clock_GHz": 1.92,
                    "bound_ms_per_step": 1e3 * bound_s, "frac": bound_s / ft,
                    "source": "profiles/r01_valu_issue_rates.txt"}
        out["fused_engine"] = {"value": relax_per_step / ft, "unit": "edge-relaxations/s",
                               "ms_per_step": 1e3 * ft, "steps": 2, "valu_roofline": valu,
                               "note": "same workload, fused engine (64 pivots per pass, bit-identical "
                                       "results, VALU-bound); not part of `value`"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rate_host, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
