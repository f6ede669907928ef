#!/bin/bash
# round 4, run 20: the two halves of a cross update on two side streams (FWX_SPLIT_CROSS, default on) against
# one after the other (=0): double-pass tests under both, then solve times by size
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py -x -q -m gpu > gpurun_out/r04_run20_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r04_run20_tests.log
for v in 0 1 0 1; do
  export FWX_SPLIT_CROSS=$v
  echo "== split cross = $v"
  python tools/measure_fused.py 6144 8192 12288 16384 --next-only --check 2>&1 | cut -c 1-200
  python tools/measure_fused.py 6144 8192 12288 16384 --rates-only --check 2>&1 | cut -c 1-200
done > gpurun_out/r04_split_cross_ab.txt 2>&1
cat gpurun_out/r04_split_cross_ab.txt | python -c "
import sys, json
cur=None
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): cur=l; print(l); continue
    try: d=json.loads(l)
    except Exception: print(l); continue
    print(d['n'], 'next' if d['next'] else 'rates', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"
