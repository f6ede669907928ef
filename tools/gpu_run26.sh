#!/bin/bash
# tracking-stage length of fused_main_arg: 16 (old), 8, 4 pivots; bits checked against the per-k engine at 8192
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run26.log
for v in 16 8 4; do
  echo sl$v >> $O/r02_run26.log
  FWX_LIB_PATH=$R/build/libfwx_sl$v.so timeout -k 10 200 python tools/measure_fused.py 8192 16384 --next-only --check >> $O/r02_run26.log 2>&1 || { tail $O/r02_run26.log; exit 1; }
  if grep -l "Memory access fault" $O/r02_run26.log 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
done
cut -c1-220 $O/r02_run26.log
