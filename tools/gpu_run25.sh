#!/bin/bash
# timing-only probes of fused_main_arg (wrong results by construction): where do the late-pass 980 us go?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo baseline > $O/r02_run25.log
timeout -k 10 120 python tools/measure_fused.py 16384 --next-only >> $O/r02_run25.log 2>&1 || exit 1
for v in 1 2 3; do
  echo probe$v >> $O/r02_run25.log
  FWX_LIB_PATH=$R/build/libfwx_probe$v.so timeout -k 10 120 python tools/measure_fused.py 16384 --next-only >> $O/r02_run25.log 2>&1 || exit 1
done
cut -c1-150 $O/r02_run25.log
