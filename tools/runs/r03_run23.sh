#!/bin/bash
# round 3, run 23: main launch against side chain with the path trace, after the arg-kernel trims (compare run 19)
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r03_prof_trace2 -o tr -- python3 $GRAFT_REPO_ROOT/tools/measure_fused.py 16384 --trace-only > $O/r03_run23.log 2>&1
grep best_ms $O/r03_run23.log | cut -c1-160
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $O/r03_prof_trace2/*/*.db $O/r03_prof_trace2/*.db 2>/dev/null | head -1) --timeline 24 --skip 1200 | cut -c1-200
