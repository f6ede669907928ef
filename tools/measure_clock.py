#!/usr/bin/env python3
"""The shader clock the chip holds while the fused main kernels run (a -DFWX_CLOCK_PROBE variant build:
python -m floydwarshall_amd.build --variant clock -DFWX_CLOCK_PROBE=1; FWX_LIB_PATH=build/variants/libfwx_clock.so).
Per solve: cycles / ticks of every main-kernel workgroup, summed -> GHz; the issue bounds of DESIGN section 4.2 are
quoted at 2.33 GHz."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from floydwarshall_amd import _lib, engine, synth  # noqa: E402

lib = ctypes.CDLL(_lib.LIB_PATH)
lib.fwx_debug_clock.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]


def clock(reset=True):
    out = (ctypes.c_ulonglong * 3)()
    assert lib.fwx_debug_clock(out, 1 if reset else 0) == 0
    return out[0], out[1], out[2]


for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384]:
    for dtype in (np.float32, np.float64):
        for with_next in (False, True):
            rate, nxt = synth.d1_uniform(n, dtype, synth.BASE_SEED + 3)
            h = engine.DeviceMatrix(n, dtype, with_next=with_next, device=0)
            best = None
            for _ in range(3):
                h.upload(rate, nxt if with_next else None)
                clock()
                t0 = time.perf_counter()
                h.solve(engine=engine.FWX_ENGINE_FUSED)
                dt = time.perf_counter() - t0
                c, r, w = clock()
                if best is None or dt < best[0]:
                    best = (dt, c, r, w)
            h.close()
            dt, c, r, w = best
            print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "next": with_next, "ms": round(1e3 * dt, 2),
                              "main_workgroups": w, "shader_clock_GHz": round(c / r * 0.1, 3),
                              "avg_workgroup_us": round(r / w / 100.0, 1)}), flush=True)
