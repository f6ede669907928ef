#!/bin/bash
# tracking-stage length 16 (built library) vs 8 after the re-scan rewrite
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo sl16 > $O/r02_run30.log
timeout -k 10 200 python tools/measure_fused.py 4096 8192 16384 --next-only >> $O/r02_run30.log 2>&1 || { tail $O/r02_run30.log; exit 1; }
echo sl8 >> $O/r02_run30.log
FWX_LIB_PATH=$R/build/libfwx_sl8.so timeout -k 10 200 python tools/measure_fused.py 4096 8192 16384 --next-only --check >> $O/r02_run30.log 2>&1 || { tail $O/r02_run30.log; exit 1; }
cut -c1-220 $O/r02_run30.log
