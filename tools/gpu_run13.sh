#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for rep in 1 2; do
for lib in cur prev; do
  echo "== $lib (rep $rep)"
  if [ $lib = prev ]; then export FWX_LIB_PATH=$R/build/libfwx_prev.so; else unset FWX_LIB_PATH; fi
  timeout -k 10 200 python tools/measure_fused.py 16384 --next-only || exit 1
  timeout -k 10 200 python tools/measure_fused.py 16384 --trace-only || exit 1
done; done > $O/r02_run13_ab.log 2>&1
grep -E "==|best_ms" $O/r02_run13_ab.log | cut -c1-160
