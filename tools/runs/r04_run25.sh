#!/bin/bash
# round 4, run 25: the 64 x 64 form of fused_main_arg runs FOUR workgroups per CU at 128 registers -- a retiring one
# frees 128 per SIMD, the panel workgroup (4 waves per SIMD x 48) needs 192: the side chain's panels starve until
# the main launch's tail (timeline of run 24).  The 128 x 64 form (three per CU at 152) leaves room.  Tile-form and
# double-pass thresholds against each other, then the timelines.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
run() { echo "== tiles_below=$1 double_pass_next_min=$2"; FWX_ARG_SMALL_TILES_BELOW=$1 FWX_DOUBLE_PASS_NEXT_MIN_N=$2 python tools/measure_fused.py 4096 5120 6144 7168 8192 10240 12288 --next-only 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  ', d['n'], d['best_ms'])
"; }
{
run 6500 8192
run 4000 8192
run 4000 6144
run 2000 6144
run 1500 5120
run 1000 4096
run 6500 8192
} 2>&1 | tee gpurun_out/r04_arg_tiles_vs_double_pass.txt
for cfg in "16384 6500" "8192 4000"; do
  set -- $cfg
  rm -rf gpurun_out/tl
  FWX_ARG_SMALL_TILES_BELOW=$2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/measure_fused.py $1 --next-only > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
  f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
  echo "== timeline N=$1 tiles_below=$2"
  python3 tools/timeline.py "$f" --dump | head -40 | tee gpurun_out/r04_timeline_$1_next_t$2.txt
  rm -rf gpurun_out/tl
done
