#!/bin/bash
# fuzz, continued: look-ahead forms forced; hostile values through every engine
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run44_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=0 FUZZ_TRAIL=$O/r02_run44_trail1.txt timeout -k 10 200 python tools/fuzz_domain.py 140 600 20261006 > $O/r02_run44_fuzz_sym.log 2>&1; rc=$?
tail -2 $O/r02_run44_fuzz_sym.log; fault; [ $rc -eq 0 ] || exit $rc
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=1000000 FUZZ_TRAIL=$O/r02_run44_trail2.txt timeout -k 10 160 python tools/fuzz_domain.py 100 600 20261007 > $O/r02_run44_fuzz_rows.log 2>&1; rc=$?
tail -2 $O/r02_run44_fuzz_rows.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/fuzz_long.py 140 400 > $O/r02_run44_fuzz_long.log 2>&1; rc=$?
tail -2 $O/r02_run44_fuzz_long.log; fault; exit $rc
