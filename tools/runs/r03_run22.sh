#!/bin/bash
# round 3, run 22: arg kernels with four-wide stage tracking and pivot search, the compaction slot in assembly, 32-bit byte addressing: parity suites, then timings
O=gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py -m gpu -x -q > $O/r03_run22_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run22_pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/measure_fused.py 16384 --check --hops 2>&1 | cut -c1-240
python tools/measure_fused.py 16384 --f64 --hops 2>&1 | cut -c1-200
python tools/measure_fused.py 32768 --next-only 2>&1 | cut -c1-200
