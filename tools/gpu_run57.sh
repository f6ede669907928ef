#!/bin/bash
# cached domain answer of a handle: the new regression test, the handle / session tests, session latency
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_session.py tests/test_gpu_multi.py -m gpu -q -x --timeout 300 > $O/r02_run57_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run57_pytest.log; [ $rc -eq 0 ] || { tail -40 $O/r02_run57_pytest.log; exit $rc; }
timeout -k 10 300 python tools/measure_session.py > $O/r02_run57_session.txt 2>&1 || { tail $O/r02_run57_session.txt; exit 1; }
grep -v amdgpu.ids $O/r02_run57_session.txt
