#!/bin/bash
# round 3, run 32: double pass with next-hops / trace / hops (arg kernels, two passes per launch): parity, then timings
O=gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py tests/test_gpu_resume.py -m gpu -x -q > $O/r03_run32_pytest.log 2>&1; rc=$?
tail -5 $O/r03_run32_pytest.log
[ $rc -ne 0 ] && exit $rc
for mode in "" "--f64"; do
  echo "== double pass $mode"; python tools/measure_fused.py 16384 $mode --check --hops 2>&1 | cut -c1-230
  echo "== single pass $mode"; FWX_DOUBLE_PASS_NEXT_MIN_N=100000000 python tools/measure_fused.py 16384 $mode --hops --next-only 2>&1 | cut -c1-200
done
python tools/measure_fused.py 32768 --next-only 2>&1 | cut -c1-200
