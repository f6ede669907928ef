#!/bin/bash
# timing-only probes of arg_rescan's global accesses (4: no store, no cnt load; 5: store without the cnt gather; 6: gather without the store)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run27.log
for v in 4 5 6; do
  echo probe$v >> $O/r02_run27.log
  FWX_LIB_PATH=$R/build/libfwx_probe$v.so timeout -k 10 120 python tools/measure_fused.py 16384 --next-only >> $O/r02_run27.log 2>&1 || { tail $O/r02_run27.log; exit 1; }
done
cut -c1-150 $O/r02_run27.log
