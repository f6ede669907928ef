#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run14_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_host_session.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "incremental or kept_input or session or requote" > $O/r02_run14_pytest.log 2>&1; rc=$?
tail -15 $O/r02_run14_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/measure_session.py > $O/r02_run14_session.log 2>&1 || { tail $O/r02_run14_session.log; exit 1; }
grep vertices $O/r02_run14_session.log; fault
