import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from floydwarshall_amd import engine, synth
for n in (4096, 16384):
    rate, nxt = synth.d1_uniform(n, np.float32, 5)
    hops = (nxt >= 0).astype(np.int32)
    for name, code in (("fused + trace + lengths", engine.FWX_ENGINE_AUTO), ("per-k", engine.FWX_ENGINE_PERK)):
        if n == 16384 and code == engine.FWX_ENGINE_PERK:
            continue
        r, nx, hp = rate.copy(), nxt.copy(), hops.copy()
        t0 = time.perf_counter()
        engine.solve(r, nx, hp, engine=code)
        print("N=%d f32 rate+next+hops through fwx_solve_f32 (PCIe included), %s: %.0f ms" % (n, name, 1e3 * (time.perf_counter() - t0)), flush=True)
