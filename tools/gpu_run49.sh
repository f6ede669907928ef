#!/bin/bash
# fuzz after fused_main_arg_f64 (system HIP runtime): default schedules, then every look-ahead form forced
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FWX_NO_TORCH=1
FUZZ_TRAIL=$O/r02_run49_trail1.txt timeout -k 10 170 python tools/fuzz_domain.py 120 700 20261008 > $O/r02_run49_a.log 2>&1; rc=$?
tail -1 $O/r02_run49_a.log | cut -c1-200; [ $rc -eq 0 ] || { tail -20 $O/r02_run49_a.log; cat $O/r02_run49_trail1.txt; exit $rc; }
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=0 FUZZ_TRAIL=$O/r02_run49_trail2.txt timeout -k 10 120 python tools/fuzz_domain.py 70 600 20261009 > $O/r02_run49_b.log 2>&1; rc=$?
tail -1 $O/r02_run49_b.log | cut -c1-200; [ $rc -eq 0 ] || { tail -20 $O/r02_run49_b.log; cat $O/r02_run49_trail2.txt; exit $rc; }
