#!/bin/bash
# round 3, run 35: crossover of the double pass with next-hops (f32): next only and with the path trace
for n in 4096 6144 8192 12288 16384; do
  for mode in "--next-only" "--trace-only"; do
    a=$(FWX_DOUBLE_PASS_NEXT_MIN_N=100000000 python tools/measure_fused.py $n $mode | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
    b=$(FWX_DOUBLE_PASS_NEXT_MIN_N=0 python tools/measure_fused.py $n $mode | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
    echo "N=$n f32 $mode: single $a ms, double $b ms"
  done
done
python tools/measure_fused.py 16384 --hops 2>&1 | cut -c1-160
python tools/measure_fused.py 32768 --next-only 2>&1 | cut -c1-160
