#!/bin/bash
# round 3, run 14: the fuzzers on the final tree, with the double pass forced at every size it applies to
O=gpurun_out
FWX_DOUBLE_PASS_MIN_N=0 FUZZ_TRAIL=$O/r03_fuzz_dp_trail.txt timeout -k 10 230 python tools/fuzz_domain.py 180 900 20261011 > $O/r03_fuzz_domain_double_pass.log 2>&1; echo "fuzz_domain (double pass forced) rc=$?"; tail -1 $O/r03_fuzz_domain_double_pass.log
FWX_DOUBLE_PASS_MIN_N=0 timeout -k 10 200 python tools/fuzz_long.py 150 > $O/r03_fuzz_long.log 2>&1; echo "fuzz_long rc=$?"; tail -1 $O/r03_fuzz_long.log
