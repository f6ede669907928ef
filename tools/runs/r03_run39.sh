#!/bin/bash
# round 3, run 39: f64 arg fold two pivots per trip, tree maxima in the re-scan: parity suites, then timings
O=gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py tests/test_gpu_resume.py -m gpu -x -q > $O/r03_run39_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run39_pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/measure_fused.py 16384 --f64 --check --hops 2>&1 | cut -c1-230
FWX_DOUBLE_PASS_NEXT_MIN_N=0 python tools/measure_fused.py 16384 --f64 --next-only 2>&1 | cut -c1-200
python tools/measure_fused.py 16384 --hops --next-only 2>&1 | cut -c1-200
timeout -k 10 130 python tools/fuzz_domain.py 80 900 20261041 2>&1 | tail -1
