#!/usr/bin/env python3
"""Fused-engine solve times through the device-resident handle (torch-free): rates only, + next-hop
matrix, + path trace, + hops.  usage: measure_fused.py [N ...]  (default 16384)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from floydwarshall_amd import engine, hip, synth  # noqa: E402


def run(n, dtype, with_next, trace, hops, reps=3, check=None):
    rate, nxt = synth.d1_uniform(n, dtype, synth.BASE_SEED + 3)
    h = engine.DeviceMatrix(n, dtype, with_next=with_next, with_hops=hops, device=0)
    if trace:
        h.enable_path_log()
    hp = (nxt >= 0).astype(np.int32) if hops else None
    times = []
    for _ in range(reps):
        h.upload(rate, nxt if with_next else None, hp)
        t0 = time.perf_counter()
        h.solve(engine=engine.FWX_ENGINE_FUSED)
        times.append(time.perf_counter() - t0)
    out = {"n": n, "dtype": np.dtype(dtype).name, "next": with_next, "trace": trace, "hops": hops,
           "ms": [round(1e3 * t, 2) for t in times], "best_ms": round(1e3 * min(times), 2),
           "relax_per_s": float(n) ** 3 / min(times)}
    if check is not None:
        r, nx, _ = h.download()
        out["rate_equal_ref"] = bool(np.array_equal(r.view(np.uint32 if dtype == np.float32 else np.uint64),
                                                    check[0].view(np.uint32 if dtype == np.float32 else np.uint64)))
        if with_next and check[1] is not None:
            out["next_equal_ref"] = bool(np.array_equal(nx, check[1]))
    h.close()
    print(json.dumps(out), flush=True)
    return out


def main():
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384]
    f64 = "--f64" in sys.argv
    for n in sizes:
        dt = np.float64 if f64 else np.float32
        ref = None
        if "--check" in sys.argv:
            # per-k engine result as the reference (itself tied to the oracle by the test-suite)
            rate, nxt = synth.d1_uniform(n, dt, synth.BASE_SEED + 3)
            engine.solve(rate, nxt, engine=engine.FWX_ENGINE_PERK)
            ref = (rate, nxt)
        if "--trace-only" in sys.argv:
            run(n, dt, True, True, False, check=ref)
            continue
        if "--next-only" not in sys.argv:
            run(n, dt, False, False, False, check=ref)
        if "--rates-only" in sys.argv:
            continue
        run(n, dt, True, False, False, check=ref)
        if "--next-only" in sys.argv:
            continue
        run(n, dt, True, True, False, check=ref)
        if "--hops" in sys.argv:
            run(n, dt, True, False, True, reps=2, check=ref)


if __name__ == "__main__":
    main()
