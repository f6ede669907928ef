#!/bin/bash
# round 3, run 15: arg kernel with straight-line staging on full tiles: parity suites, A/B; session with odd orders
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_host_session.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py -m gpu -x -q > $O/r03_run15_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run15_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "--- general staging"; FWX_ARG_GENERAL_STAGING=1 python tools/measure_fused.py 16384 --check --next-only 2>&1 | cut -c1-230
echo "--- straight-line staging"; python tools/measure_fused.py 16384 --check --next-only 2>&1 | cut -c1-230
echo "--- N=8192"; FWX_ARG_GENERAL_STAGING=1 python tools/measure_fused.py 8192 --next-only 2>&1 | cut -c1-200; python tools/measure_fused.py 8192 --next-only 2>&1 | cut -c1-200
python tools/measure_session.py > $O/r03_session_latency.txt 2>&1; cat $O/r03_session_latency.txt | cut -c1-330
