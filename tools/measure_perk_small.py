#!/usr/bin/env python3
"""BASELINE config 2 (N = 1024 f64, per-k engine: 1024 launches of a cache-resident 8 MiB matrix) is
bound by the launch path, not by the GPU.  This measures it on BOTH HIP runtimes a Python caller can
end up on: the one libfwx is built against (/opt/rocm; default) and, with --torch, the one bundled
with the torch wheel (imported first, as rounds 1-2 did).  Per solve and per sixteenth of the pivots.
usage: measure_perk_small.py [--torch] [n]"""
import json
import sys
import time

import numpy as np

if "--torch" in sys.argv:
    import torch  # noqa: F401  (first: libfwx then binds to torch's bundled libamdhip64)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import warnings  # noqa: E402
warnings.simplefilter("ignore", RuntimeWarning)
from floydwarshall_amd import _lib, engine, hip, synth  # noqa: E402

n = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 1024
rate64, _ = synth.GENERATORS["d1"](n, np.float64, synth.BASE_SEED + 1)
hip.set_device(0)
st = hip.Stream()
pristine = hip.DeviceArray.from_numpy(rate64)
rate = hip.DeviceArray(pristine.shape, np.float64)
SEG = 16
res = []
for rep in range(6):
    rate.copy_(pristine, st)
    st.synchronize()
    evs = []
    t0 = time.perf_counter()
    for a, b in zip([n * i // SEG for i in range(SEG)], [n * (i + 1) // SEG for i in range(SEG)]):
        evs.append(hip.Event())
        evs[-1].record(st)
        engine.dev_relax(rate, n, 0, a, b, stream=st)
    evs.append(hip.Event())
    evs[-1].record(st)
    t_host = time.perf_counter() - t0           # all launches enqueued
    st.synchronize()
    t_all = time.perf_counter() - t0
    seg = [round(1e3 * evs[i].elapsed_time(evs[i + 1]) / (n // SEG), 2) for i in range(SEG)]
    res.append({"ms": round(1e3 * t_all, 3), "host_enqueue_ms": round(1e3 * t_host, 3), "us_per_launch_by_sixteenth": seg})
built, run, same = _lib.runtime_versions()
print(json.dumps({"n": n, "dtype": "f64", "engine": "per-k", "hip_built_against": built, "hip_runtime": run,
                  "torch_first": "--torch" in sys.argv, "best_ms": min(r["ms"] for r in res[1:]), "solves": res[1:]}))
