#!/bin/bash
# round 3, run 29: f64 arg kernel on 32 x 64 tiles (three workgroups per CU) against 64 x 64 (two): parity, A/B timings
O=$PWD/gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py tests/test_gpu_resume.py -m gpu -x -q > $O/r03_run29_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run29_pytest.log
[ $rc -ne 0 ] && exit $rc
echo "== 32 x 64 (default)"; python tools/measure_fused.py 16384 --f64 --hops 2>&1 | cut -c1-200
echo "== 64 x 64"; FWX_ARG_F64_TALL_TILES=1 python tools/measure_fused.py 16384 --f64 --hops 2>&1 | cut -c1-200
for n in 2048 4096 8192; do
  a=$(python tools/measure_fused.py $n --f64 --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  b=$(FWX_ARG_F64_TALL_TILES=1 python tools/measure_fused.py $n --f64 --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  echo "N=$n f64+next: 32x64 $a ms, 64x64 $b ms"
done
