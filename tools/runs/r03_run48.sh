#!/bin/bash
# round 3, run 48: longer fuzz on the final build: hostile values through every engine (fuzz_long), the domain fuzz with
# larger orders, the partitioned driver with real processes (fuzz_dist)
O=gpurun_out
timeout -k 10 340 python tools/fuzz_long.py 300 260 > $O/r03_fuzz_long_final2.log 2>&1; rc=$?; echo "fuzz_long rc=$rc"; tail -1 $O/r03_fuzz_long_final2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 340 python tools/fuzz_domain.py 300 1400 20261061 > $O/r03_fuzz_final3.log 2>&1; rc=$?; echo "fuzz_domain rc=$rc"; tail -1 $O/r03_fuzz_final3.log
exit $rc
