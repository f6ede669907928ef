#!/bin/bash
# round 3, run 1: the whole -m gpu suite on the torch-free binding, then the N > 1 bench rehearsal
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_run01_pytest.log 2>&1; rc=$?
tail -5 $O/r03_run01_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --devices 0,0 --steps 1 --warmup 1 > $O/r03_run01_bench_p2.json 2> $O/r03_run01_bench_p2.err; rc=$?
tail -c 1500 $O/r03_run01_bench_p2.json; tail -3 $O/r03_run01_bench_p2.err
exit $rc
