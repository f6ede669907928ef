#!/bin/bash
# round 3, run 36: parity suites on the two-pass arg kernels (two inlined copies of the pass body)
O=gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py tests/test_gpu_resume.py tests/test_gpu_host_session.py -m gpu -x -q > $O/r03_run36_pytest.log 2>&1; rc=$?
tail -4 $O/r03_run36_pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/measure_fused.py 16384 --check --hops 2>&1 | cut -c1-250
FWX_DOUBLE_PASS_NEXT_MIN_N=0 timeout -k 10 160 python tools/fuzz_domain.py 100 900 20261031 2>&1 | tail -1
