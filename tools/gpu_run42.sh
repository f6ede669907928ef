#!/bin/bash
# fuzz of the final code: default schedules (serial, merged panels, AUTO fused from n = 65), then every
# look-ahead form forced; hostile values through every engine; concurrent per-call solves (context pool)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run42_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
FUZZ_TRAIL=$O/r02_run42_trail1.txt timeout -k 10 260 python tools/fuzz_domain.py 200 700 20261005 > $O/r02_run42_fuzz_default.log 2>&1; rc=$?
tail -2 $O/r02_run42_fuzz_default.log; fault; [ $rc -eq 0 ] || exit $rc
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=0 timeout -k 10 200 python tools/fuzz_domain.py 140 600 20261006 > $O/r02_run42_fuzz_sym.log 2>&1; rc=$?
tail -2 $O/r02_run42_fuzz_sym.log; fault; [ $rc -eq 0 ] || exit $rc
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=1000000 timeout -k 10 160 python tools/fuzz_domain.py 100 600 20261007 > $O/r02_run42_fuzz_rows.log 2>&1; rc=$?
tail -2 $O/r02_run42_fuzz_rows.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/fuzz_long.py 140 400 > $O/r02_run42_fuzz_long.log 2>&1; rc=$?
tail -2 $O/r02_run42_fuzz_long.log; fault; exit $rc
