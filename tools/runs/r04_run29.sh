#!/bin/bash
# round 4, run 29: serial against symmetric look-ahead again (the crossovers in fwx_api.hip date from round 2, before the
# 48-register panels and the side chain's wave priority): FWX_LOOKAHEAD_MIN_N=0 forces the look-ahead
cd "$GRAFT_REPO_ROOT"
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'])
"; }
for la in default 0 default 0; do
  if [ $la = default ]; then unset FWX_LOOKAHEAD_MIN_N; else export FWX_LOOKAHEAD_MIN_N=$la; fi
  echo "== FWX_LOOKAHEAD_MIN_N=$la"
  one 2048 3072 4096 5120 6144 7168 --next-only
  one 2048 4096 6144 --trace-only
  one 2048 4096 --rates-only
  one 2048 4096 6144 8192 12288 --f64 --next-only
  one 4096 8192 --f64 --rates-only
done 2>&1 | tee gpurun_out/r04_lookahead_crossover.txt
