#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 300 python tools/measure_fused.py 4096 16384 --check --next-only > $O/r02_arg2.log 2>&1 || { tail -20 $O/r02_arg2.log; exit 1; }
cat $O/r02_arg2.log | cut -c1-260
for v in 0 9; do echo "f64 variant $v"; FWX_MAXF64_VARIANT=$v timeout -k 10 200 python tools/measure_fused.py 4096 16384 --f64 --rates-only --check || exit 1; done > $O/r02_f64_ab.log 2>&1
grep -E "variant|best_ms" $O/r02_f64_ab.log | cut -c1-260
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py -m gpu -q -x --timeout 600 -k "fused or max_form or exact or multi or partition or whole or config or hostile or hops" > $O/r02_pytest3.log 2>&1
tail -6 $O/r02_pytest3.log
