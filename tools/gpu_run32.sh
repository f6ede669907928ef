#!/bin/bash
# kernel trace of small / mid sizes (panel-chain bound): N=1024, 2048, 4096 f32 rates + next
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_mid -o f --output-format csv -- python3 $R/tools/measure_fused.py 1024 2048 4096 --next-only > $O/r02_run32.log 2>&1 || { tail $O/r02_run32.log; exit 1; }
cut -c1-160 $O/r02_run32.log
