#!/bin/bash
# round 4, run 35: same-box A/B of the panel chain's synchronisation: barrier per sub-block (variant barrier) / flag per
# pivot (shipped) / flags + wave priorities by phase (variant flagprio)
cd "$GRAFT_REPO_ROOT"
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'])
"; }
for v in barrier flags flagprio barrier flags flagprio; do
  case $v in barrier) export FWX_LIB_PATH=$PWD/build/variants/libfwx_barrier.so;; flags) unset FWX_LIB_PATH;; flagprio) export FWX_LIB_PATH=$PWD/build/variants/libfwx_flagprio.so;; esac
  echo "== $v"
  one 512 1024 2048 4096 8192 16384 --rates-only
  one 512 1024 2048 4096 6144 8192 16384 --next-only
  one 1024 2048 4096 --f64 --next-only
  one 1024 --f64 --trace-only
done 2>&1 | tee gpurun_out/r04_panel_flags_ab.txt
