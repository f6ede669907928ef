#!/bin/bash
# whole-oracle parity at N=16384 for the final code (interior path + symmetric look-ahead): rates, then + next
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 560 python3 tools/full_parity_n16384.py $O/r02_full_parity_n16384.json > $O/r02_run24_a.log 2>&1; rc=$?
tail -4 $O/r02_run24_a.log; [ $rc -eq 0 ] || exit $rc
if grep -l "Memory access fault" $O/r02_run24_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
timeout -k 10 560 python3 tools/full_parity_n16384.py $O/r02_full_parity_n16384_next.json 16384 --next > $O/r02_run24_b.log 2>&1; rc=$?
tail -4 $O/r02_run24_b.log; exit $rc
