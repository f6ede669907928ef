#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in 0 1 2 3 9; do echo "f64 variant $v"; FWX_MAXF64_VARIANT=$v timeout -k 10 200 python tools/measure_fused.py 16384 --f64 --rates-only --check || exit 1; done > $O/r02_run8_f64.log 2>&1
grep -E "variant|best_ms" $O/r02_run8_f64.log | cut -c1-230
if grep -l "Memory access fault" $O/r02_run8_*.log 2>/dev/null; then exit 9; fi
