#!/bin/bash
# round 3, run 19: main launch against side chain, N = 16384 with the path trace and with hops
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r03_prof_trace -o tr -- python3 $GRAFT_REPO_ROOT/tools/measure_fused.py 16384 --trace-only > $O/r03_run19.log 2>&1
grep best_ms $O/r03_run19.log | cut -c1-160
