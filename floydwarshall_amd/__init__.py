"""floydwarshall_amd -- MI355X-native (gfx950) max-product Floyd-Warshall engine.

Drop-in for ONE path of jinilover/floydWarshall: the k-i-j relaxation `runAlgo` behind
`floydWarshall` (/root/reference/src/lib/Algorithms.hs:19-20, :42-61).  The compute lives in
libfwx.so (hand-written HIP, C ABI in include/fwx.h); this package is the thin host plumbing over
it.  There is no CPU fallback.
"""
from .engine import (DeviceMatrix, FwxError, dev_panel, dev_relax, device_count, follow_path,  # noqa: F401
                     solve)

__version__ = "0.1.0"
