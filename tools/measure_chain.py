import os, sys, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from floydwarshall_amd import engine, synth
for n in (6144, 8192, 12288):
    rate, nxt = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 2)
    for with_next in (False, True):
        h = engine.DeviceMatrix(n, np.float32, with_next=with_next, devices=[0])
        ts = []
        for _ in range(3):
            h.upload(rate, nxt if with_next else None); t0 = time.perf_counter(); h.solve(); ts.append(time.perf_counter() - t0)
        h.set_timing(True); h.upload(rate, nxt if with_next else None); h.solve(); t = h.timing(); h.close()
        s = engine.DeviceMatrix(n, np.float32, with_next=with_next, device=0)
        t2 = []
        for _ in range(3):
            s.upload(rate, nxt if with_next else None); t0 = time.perf_counter(); s.solve(); t2.append(time.perf_counter() - t0)
        s.close()
        print(json.dumps({"n": n, "next": with_next, "multi_P1_ms": round(1e3 * min(ts), 2), "single_ms": round(1e3 * min(t2), 2),
                          "pivots_per_step": t["pivots_per_step"], "bulk_us": round(t["bulk_us"], 1), "chain_us": round(t["chain_us"], 1),
                          "panel_us": round(t["panel_us"], 1)}), flush=True)
