#!/bin/bash
# system HIP runtime (FWX_NO_TORCH=1), default schedules, then the rows-only look-ahead forced
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FWX_NO_TORCH=1
FUZZ_TRAIL=$O/r02_run46_trail1.txt timeout -k 10 220 python tools/fuzz_domain.py 170 700 20261005 > $O/r02_run46_fuzz_default.log 2>&1; rc=$?
tail -1 $O/r02_run46_fuzz_default.log | cut -c1-200; [ $rc -eq 0 ] || { cat $O/r02_run46_trail1.txt; exit $rc; }
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=1000000 FUZZ_TRAIL=$O/r02_run46_trail2.txt timeout -k 10 130 python tools/fuzz_domain.py 80 600 20261007 > $O/r02_run46_fuzz_rows.log 2>&1; rc=$?
tail -1 $O/r02_run46_fuzz_rows.log | cut -c1-200; [ $rc -eq 0 ] || { cat $O/r02_run46_trail2.txt; exit $rc; }
