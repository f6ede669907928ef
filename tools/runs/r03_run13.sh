#!/bin/bash
# round 3, run 13: kernel trace of the lazy next-hop solve at N = 8192
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
export FWX_LAZY_NEXT_MIN_N=0
rocprofv3 --kernel-trace --stats -d $O/r03_prof_lazy -o lz -- python3 $GRAFT_REPO_ROOT/tools/measure_fused.py 8192 --next-only > $O/r03_run13.log 2>&1
grep best_ms $O/r03_run13.log | cut -c1-160
