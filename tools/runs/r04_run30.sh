#!/bin/bash
# round 4, run 30: EXPERIMENT -- per-iteration wave priority inside the panel kernels (the wave on the serial phase and
# the one that goes serial next at 3, the others' apply phases at 1): variant dynprio against the shipped build
cd "$GRAFT_REPO_ROOT"
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
for v in base dyn base dyn; do
  if [ $v = base ]; then unset FWX_LIB_PATH; else export FWX_LIB_PATH=$PWD/build/variants/libfwx_dynprio.so; fi
  echo "== $v"
  one 512 1024 2048 4096 8192 --rates-only --check
  one 1024 2048 4096 6144 --next-only --check
  one 1024 2048 4096 --f64 --next-only --check
  one 1024 4096 --f64 --trace-only
done 2>&1 | tee gpurun_out/r04_panel_dynprio.txt
