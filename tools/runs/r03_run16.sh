#!/bin/bash
# round 3, run 16: straight-line staging in both arg kernels: parity suites, then the field-set table and config 5
O=gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_resume.py tests/test_gpu_multi.py -m gpu -x -q > $O/r03_run16_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run16_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "--- f64 general staging"; FWX_ARG_GENERAL_STAGING=1 python tools/measure_fused.py 16384 --f64 --next-only 2>&1 | cut -c1-200
echo "--- f64 straight-line"; python tools/measure_fused.py 16384 --f64 --next-only 2>&1 | cut -c1-200
python tools/measure_fused.py 16384 --hops > $O/r03_fused_n16384_f32.jsonl 2>&1; cat $O/r03_fused_n16384_f32.jsonl | cut -c1-200
python tools/measure_fused.py 16384 --f64 --hops > $O/r03_fused_n16384_f64.jsonl 2>&1; cat $O/r03_fused_n16384_f64.jsonl | cut -c1-200
python tools/measure_fused.py 32768 --next-only > $O/r03_fused_n32768_f32.jsonl 2>&1; cat $O/r03_fused_n32768_f32.jsonl | cut -c1-200
python tools/measure_fused.py 8192 4096 1024 > $O/r03_fused_small_f32.jsonl 2>&1; cat $O/r03_fused_small_f32.jsonl | cut -c1-200
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline --no-extras > $O/r03_bench_config5.json 2> $O/r03_bench_config5.err; echo "config 5 rc=$?"
