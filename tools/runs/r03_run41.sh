#!/bin/bash
# round 3, run 41: kernel traces of the final fused solves at N = 16384: rates + next (two passes per launch), with the
# path trace, f64 + next, and rates only
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tag in next trace f64next rates; do
  case $tag in next) a="--next-only";; trace) a="--trace-only";; f64next) a="--f64 --next-only";; rates) a="--rates-only";; esac
  rocprofv3 --kernel-trace --stats -d $O/r03_final_prof_$tag -o p -- python3 $R/tools/measure_fused.py 16384 $a > $O/r03_run41_$tag.log 2>&1 || exit 1
  echo "== $tag: $(grep -o '"best_ms": [0-9.]*' $O/r03_run41_$tag.log)"
  python3 $R/tools/rocpd_summary.py $(ls $O/r03_final_prof_$tag/*.db $O/r03_final_prof_$tag/*/*.db 2>/dev/null | head -1) | head -7 | cut -c1-150
done
