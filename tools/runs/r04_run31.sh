#!/bin/bash
# round 4, run 31: the look-ahead chain of the double pass as one panel launch per block (pre-apply): schedule tests, then
# FWX_PANEL_PRE=0/1 by size, with the double pass forced from N = 2048
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py -x -q -m gpu > gpurun_out/r04_run31_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_run31_tests.log
[ $rc -eq 0 ] || exit $rc
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
for cfg in "1 default" "0 default" "1 0" "0 0" "1 default" ; do
  set -- $cfg
  export FWX_PANEL_PRE=$1
  if [ $2 = default ]; then unset FWX_DOUBLE_PASS_MIN_N FWX_DOUBLE_PASS_NEXT_MIN_N; else export FWX_DOUBLE_PASS_MIN_N=$2 FWX_DOUBLE_PASS_NEXT_MIN_N=$2; fi
  echo "== pre=$1 double_pass_min=$2"
  one 2048 3072 4096 6144 8192 16384 --rates-only --check
  one 2048 3072 4096 6144 8192 16384 --next-only --check
  one 4096 8192 --f64 --rates-only --check
done 2>&1 | tee gpurun_out/r04_panel_pre.txt
