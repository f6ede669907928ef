import time, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np
from floydwarshall_amd import engine, synth
import helpers
T=[time.perf_counter()]
def lap(msg):
    T.append(time.perf_counter()); print("%-40s %.1f s" % (msg, T[-1]-T[-2]), flush=True)
n, P = 32768, 8
rate0, next0 = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 4); lap("generate input")
hops0 = (next0 >= 0).astype(np.int32); lap("hops0")
dm = engine.DeviceMatrix(n, np.float32, with_next=True, with_hops=True, devices=[0]*P); lap("create")
dm.upload(rate0, next0, hops0); lap("upload 12 GiB")
dm.solve(k_begin=0, k_end=256); lap("solve 256 pivots")
a = dm.download(); lap("download 12 GiB")
d = [helpers.digest(x) for x in a]; lap("3 digests")
helpers.assert_bits_equal(a[0], a[0]); lap("one compare 4 GiB")
dm.solve(k_begin=256, k_end=16256); lap("solve 16000 pivots")
dm.close(); lap("close")
