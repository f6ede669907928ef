#!/bin/bash
# round 3, run 12: lazy next-hops: parity at forced small sizes, then A/B at N = 16384 / 8192 / 32768
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lazy_next.py -m gpu -x -q > $O/r03_run12_pytest.log 2>&1; rc=$?
tail -5 $O/r03_run12_pytest.log
[ $rc -ne 0 ] && exit $rc
for n in 16384 8192; do
echo "--- N=$n arg kernels"; FWX_LAZY_NEXT_MIN_N=100000000 timeout -k 10 300 python tools/measure_fused.py $n --check --next-only 2>&1 | tee -a $O/r03_run12_arg.log
echo "--- N=$n lazy"; FWX_LAZY_NEXT_MIN_N=0 timeout -k 10 300 python tools/measure_fused.py $n --check --next-only 2>&1 | tee -a $O/r03_run12_lazy.log
done
echo "--- N=32768 lazy"; FWX_LAZY_NEXT_MIN_N=0 timeout -k 10 300 python tools/measure_fused.py 32768 --next-only 2>&1 | tee -a $O/r03_run12_lazy.log
echo "--- f64 N=16384 arg / lazy"; FWX_LAZY_NEXT_MIN_N=100000000 timeout -k 10 300 python tools/measure_fused.py 16384 --f64 --next-only 2>&1 | tee -a $O/r03_run12_arg.log
FWX_LAZY_NEXT_MIN_N=0 timeout -k 10 300 python tools/measure_fused.py 16384 --f64 --next-only 2>&1 | tee -a $O/r03_run12_lazy.log
