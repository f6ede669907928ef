#!/bin/bash
# round 3, run 6: the double-pass schedule: parity at forced small sizes, then A/B at N = 16384 (f32, f64), 8192, 12288
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_double_pass.py -m gpu -x -q > $O/r03_run06_pytest.log 2>&1; rc=$?
tail -5 $O/r03_run06_pytest.log
[ $rc -ne 0 ] && exit $rc
for n in 16384 12288 8192; do
echo "--- N=$n single pass"; FWX_DOUBLE_PASS_MIN_N=100000000 timeout -k 10 300 python tools/measure_fused.py $n --check --rates-only 2>&1 | tee -a $O/r03_run06_single.log
echo "--- N=$n double pass"; FWX_DOUBLE_PASS_MIN_N=0 timeout -k 10 300 python tools/measure_fused.py $n --check --rates-only 2>&1 | tee -a $O/r03_run06_double.log
done
echo "--- f64 N=16384 single"; FWX_DOUBLE_PASS_MIN_N=100000000 timeout -k 10 300 python tools/measure_fused.py 16384 --f64 --rates-only 2>&1 | tee -a $O/r03_run06_single.log
echo "--- f64 N=16384 double"; FWX_DOUBLE_PASS_MIN_N=0 timeout -k 10 300 python tools/measure_fused.py 16384 --f64 --rates-only 2>&1 | tee -a $O/r03_run06_double.log
