#!/bin/bash
# round 3, run 7: kernel trace of the double-pass schedule at N = 16384 (rates only)
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
export FWX_DOUBLE_PASS_MIN_N=0
rocprofv3 --kernel-trace --stats -d $O/r03_prof_double -o dp -- python3 $GRAFT_REPO_ROOT/tools/measure_fused.py 16384 --rates-only > $O/r03_run07.log 2>&1
tail -3 $O/r03_run07.log
find $O/r03_prof_double -name "*stats*" | head
