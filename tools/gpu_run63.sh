#!/bin/bash
# whole-oracle parity at N=16384 with next-hops on the round's final code
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 480 python3 tools/full_parity_n16384.py $O/r02_full_parity_n16384_next.json 16384 --next > $O/r02_run63.log 2>&1; rc=$?
tail -4 $O/r02_run63.log | cut -c1-400; exit $rc
