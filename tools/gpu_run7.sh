#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run7_*.log 2>/dev/null; then echo "GPU FAULT in the logs above"; exit 9; fi; }
for v in 0 9; do echo "f64 variant $v"; FWX_MAXF64_VARIANT=$v timeout -k 10 200 python tools/measure_fused.py 4096 16384 --f64 --rates-only --check || exit 1; done > $O/r02_run7_f64.log 2>&1
grep -E "variant|best_ms" $O/r02_run7_f64.log | cut -c1-230; fault
timeout -k 10 600 python -m pytest tests/test_gpu_host_session.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "session or overlap or max_form or fused_engine or hostile or full_size" > $O/r02_run7_pytest.log 2>&1; rc=$?
tail -6 $O/r02_run7_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/measure_fused.py 32768 --next-only > $O/r02_run7_cfg5.log 2>&1 || exit 1
cut -c1-230 $O/r02_run7_cfg5.log; fault
timeout -k 10 900 python tools/full_parity_n16384.py profiles/r02_full_parity_n16384.json > $O/r02_run7_parity.log 2>&1; rc=$?
tail -4 $O/r02_run7_parity.log | cut -c1-600; fault
cp profiles/r02_full_parity_n16384.json $O/ 2>/dev/null
exit $rc
