#!/bin/bash
# round 4, run 40: EXPERIMENT -- is it LDS fragmentation that keeps the panels out beside the 64 x 64 fused_main_arg?
# 12.25 KB of unused dynamic LDS per main workgroup (three per CU, 48.75 KB holes) with and without the split main launch
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
run() { echo "== pad=$1 split=$2"; FWX_ARG_PAD_LDS=$1 FWX_SPLIT_MAIN=$2 python tools/measure_fused.py 6144 8192 9216 10240 --next-only --check 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
"; }
{ run 0 1; run 12544 0; run 12544 1; run 0 0; run 0 1; run 12544 0; } 2>&1 | tee gpurun_out/r04_arg_pad_lds.txt
rm -rf gpurun_out/tl
FWX_ARG_PAD_LDS=12544 FWX_SPLIT_MAIN=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/measure_fused.py 8192 --next-only > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$f" --dump > gpurun_out/r04_timeline_8192_next_padlds.txt; sed -n 1,24p gpurun_out/r04_timeline_8192_next_padlds.txt
rm -rf gpurun_out/tl
