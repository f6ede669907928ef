#!/bin/bash
# round 3, run 5: the 128 x 128 arg kernel: parity at forced small sizes, then A/B at N = 16384 / 32768
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_arg_wide.py -m gpu -x -q > $O/r03_run05_pytest.log 2>&1; rc=$?
tail -5 $O/r03_run05_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "--- 128x64 (wide off)"; FWX_ARG_WIDE_MIN_TILES=100000000 timeout -k 10 300 python tools/measure_fused.py 16384 --check --hops 2>&1 | tee $O/r03_run05_narrow.log
echo "--- 128x128 (default)"; timeout -k 10 300 python tools/measure_fused.py 16384 --check --hops 2>&1 | tee $O/r03_run05_wide.log
echo "--- config 5 size"; FWX_ARG_WIDE_MIN_TILES=100000000 timeout -k 10 300 python tools/measure_fused.py 32768 --next-only 2>&1 | tee $O/r03_run05_n32768_narrow.log
timeout -k 10 300 python tools/measure_fused.py 32768 --next-only 2>&1 | tee $O/r03_run05_n32768_wide.log
