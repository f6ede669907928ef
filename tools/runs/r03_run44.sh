#!/bin/bash
# round 3, run 44: domain fuzz on the final build (48-register panels, two-pass arg kernels, f64 two pivots per trip),
# default thresholds and with the double pass forced at every order
O=gpurun_out
timeout -k 10 200 python tools/fuzz_domain.py 150 900 20261051 > $O/r03_fuzz_final2.log 2>&1; rc=$?; echo "fuzz_domain rc=$rc"; tail -1 $O/r03_fuzz_final2.log
[ $rc -ne 0 ] && exit $rc
FWX_DOUBLE_PASS_NEXT_MIN_N=0 FWX_DOUBLE_PASS_MIN_N=0 timeout -k 10 200 python tools/fuzz_domain.py 150 900 20261052 > $O/r03_fuzz_final2_dp.log 2>&1; rc=$?; echo "fuzz_domain (double pass forced) rc=$rc"; tail -1 $O/r03_fuzz_final2_dp.log
exit $rc
