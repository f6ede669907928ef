#!/bin/bash
# round 4, run 32: EXPERIMENT -- single-pass look-ahead whose chain is ONE panel launch with the previous pass pre-applied
# (FWX_PANEL_PRE1=1, look-ahead forced with FWX_LOOKAHEAD_MIN_N=0) against the serial schedule
cd "$GRAFT_REPO_ROOT"
FWX_PANEL_PRE1=1 timeout -k 10 400 python -m pytest tests/test_gpu_symmetric.py -x -q -m gpu > gpurun_out/r04_run32_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_run32_tests.log
[ $rc -eq 0 ] || exit $rc
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
for cfg in serial pre1 serial pre1; do
  if [ $cfg = serial ]; then unset FWX_PANEL_PRE1 FWX_LOOKAHEAD_MIN_N; else export FWX_PANEL_PRE1=1 FWX_LOOKAHEAD_MIN_N=0; fi
  echo "== $cfg"
  one 1024 2048 3072 4096 5120 --rates-only --check
  one 1024 2048 3072 4096 6144 7168 --next-only --check
  one 2048 4096 8192 --f64 --next-only --check
  one 2048 4096 --f64 --rates-only --check
done 2>&1 | tee gpurun_out/r04_panel_pre1.txt
