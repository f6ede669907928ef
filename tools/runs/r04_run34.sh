#!/bin/bash
# round 4, run 34: panel chain synchronised by one LDS flag per pivot instead of a workgroup barrier per sub-block
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_double_pass.py -x -q -m gpu > gpurun_out/r04_run34_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_run34_tests.log
[ $rc -eq 0 ] || exit $rc
one() { python tools/measure_fused.py "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
{
one 256 512 1024 2048 4096 6144 8192 16384 --rates-only --check
one 256 512 1024 2048 4096 6144 8192 16384 --next-only --check
one 1024 4096 --trace-only
one 512 1024 2048 4096 --f64 --next-only --check
one 1024 4096 --f64 --rates-only --check
} 2>&1 | tee gpurun_out/r04_panel_flags.txt
