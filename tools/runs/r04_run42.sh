#!/bin/bash
# round 4, run 42: the double-pass crossovers again (panels by flags, 32-row column workgroups beside the 64 x 64 arg form,
# side-chain priorities): FWX_DOUBLE_PASS_NEXT_MIN_N / FWX_DOUBLE_PASS_MIN_N forced lower against the defaults
cd "$GRAFT_REPO_ROOT"
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'])
"; }
for cfg in default low default low; do
  if [ $cfg = default ]; then unset FWX_DOUBLE_PASS_NEXT_MIN_N FWX_DOUBLE_PASS_MIN_N; else export FWX_DOUBLE_PASS_NEXT_MIN_N=3072 FWX_DOUBLE_PASS_MIN_N=3072; fi
  echo "== $cfg"
  one 3072 4096 5120 6144 7168 --next-only
  one 4096 6144 7168 --trace-only
  one 3072 4096 5120 --rates-only
done 2>&1 | tee gpurun_out/r04_double_pass_crossover.txt
