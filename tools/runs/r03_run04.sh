#!/bin/bash
# round 3, run 4: resume tests, the session tests again, session latency table
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_resume.py tests/test_gpu_host_session.py -m gpu -x -q > $O/r03_run04_pytest.log 2>&1; rc=$?
tail -5 $O/r03_run04_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/measure_session.py > $O/r03_session_latency.txt 2>&1; rc=$?
cat $O/r03_session_latency.txt
exit $rc
