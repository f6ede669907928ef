#!/bin/bash
# round 4, run 37: snapshot stores after the serial phase's four steps instead of inside them: tests, A/B against the previous build
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r04_run37_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_run37_tests.log
[ $rc -eq 0 ] || exit $rc
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
for v in prev new prev new; do
  case $v in prev) export FWX_LIB_PATH=$PWD/build/variants/libfwx_prev.so;; new) unset FWX_LIB_PATH;; esac
  echo "== $v"
  one 256 512 1024 2048 4096 --rates-only --check
  one 512 1024 2048 4096 8192 --next-only --check
  one 1024 --trace-only
  one 1024 2048 --f64 --next-only --check
  one 1024 2048 --f64 --rates-only --check
done 2>&1 | tee gpurun_out/r04_panel_late_stores_ab.txt
