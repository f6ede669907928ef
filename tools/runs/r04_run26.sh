#!/bin/bash
# round 4, run 26: EXPERIMENT -- the main launch of a pair as TWO launches (rows [0, h), [h, n)): the first one's tail
# lets the starving panel workgroups of the side chain in (see run 24/25)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
run() { echo "== split=$1 double_pass_next_min=$2"; FWX_SPLIT_MAIN=$1 FWX_DOUBLE_PASS_NEXT_MIN_N=$2 python tools/measure_fused.py 5120 6144 7168 8192 10240 12288 --next-only --check 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
{
run 0 8192
run 50 8192
run 35 8192
run 65 8192
run 50 5120
run 35 5120
run 0 8192
} 2>&1 | tee gpurun_out/r04_split_main.txt
rm -rf gpurun_out/tl
FWX_SPLIT_MAIN=50 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/measure_fused.py 8192 --next-only > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$f" --dump > gpurun_out/r04_timeline_8192_next_split50.txt; sed -n 1,45p gpurun_out/r04_timeline_8192_next_split50.txt
rm -rf gpurun_out/tl
