#!/bin/bash
# round 3, run 18: per-pass durations of the arg main kernel after this round's changes (N = 16384, rates + next)
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r03_prof_arg -o arg -- python3 $GRAFT_REPO_ROOT/tools/measure_fused.py 16384 --next-only > $O/r03_run18.log 2>&1
grep best_ms $O/r03_run18.log | cut -c1-160
