#!/bin/bash
# pinned staging of small host transfers in fwx_solve_*: host-API parity tests, then call latency
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_inputs.py tests/test_gpu_symmetric.py tests/test_c_consumer.py -m gpu -q -x --timeout 600 > $O/r02_run53_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run53_pytest.log; [ $rc -eq 0 ] || exit $rc
if grep -l "Memory access fault" $O/r02_run53_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
timeout -k 10 300 python tools/measure_call_latency.py > $O/r02_call_latency_staged.txt 2>&1 || { tail $O/r02_call_latency_staged.txt; exit 1; }
cat $O/r02_call_latency_staged.txt
