#!/bin/bash
# round 4, run 41: column panel workgroups of 32 rows (32.25 KB of LDS) beside the 64 x 64 fused_main_arg: schedule tests, then
# A/B (32-row form on/off x split main launch on/off) and the timeline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py -x -q -m gpu > gpurun_out/r04_run41_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_run41_tests.log
[ $rc -eq 0 ] || exit $rc
run() { echo "== rows32=$1 split=$2"; FWX_PANELS_32_ROWS=$1 FWX_SPLIT_MAIN=$2 python tools/measure_fused.py 8192 9216 10240 --next-only --check 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
{ run 0 1; run 1 0; run 1 1; run 0 1; run 1 0; run 1 1; } 2>&1 | tee gpurun_out/r04_panels_32_rows.txt
rm -rf gpurun_out/tl
FWX_SPLIT_MAIN=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/measure_fused.py 8192 --next-only > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$f" --dump > gpurun_out/r04_timeline_8192_next_rows32.txt; sed -n 1,22p gpurun_out/r04_timeline_8192_next_rows32.txt
rm -rf gpurun_out/tl
