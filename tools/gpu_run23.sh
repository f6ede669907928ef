#!/bin/bash
# fuzz with the symmetric look-ahead forced at every size (FWX_SYMMETRIC_MIN_N=0): domain fuzz, then hostile values
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FWX_SYMMETRIC_MIN_N=0
FUZZ_TRAIL=$O/r02_run23_trail1.txt timeout -k 10 330 python tools/fuzz_domain.py 270 700 20261004 > $O/r02_run23_fuzz_domain.log 2>&1; rc=$?
tail -3 $O/r02_run23_fuzz_domain.log
if grep -l "Memory access fault" $O/r02_run23_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/fuzz_long.py 240 > $O/r02_run23_fuzz_long.log 2>&1; rc=$?
tail -3 $O/r02_run23_fuzz_long.log
if grep -l "Memory access fault" $O/r02_run23_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
exit $rc
