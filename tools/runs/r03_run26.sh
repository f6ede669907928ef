#!/bin/bash
# round 3, run 26: 64 x 64 against 128 x 64 tiles of the f32 arg kernel at large orders, after the trims
for n in 8192 12288 16384; do
  a=$(python tools/measure_fused.py $n --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  b=$(FWX_ARG_SMALL_TILES_BELOW=100000000 python tools/measure_fused.py $n --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  c=$(FWX_ARG_SMALL_TILES_BELOW=0 python tools/measure_fused.py $n --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  echo "N=$n f32+next: default $a ms, 64x64 tiles $b ms, 128x64 tiles $c ms"
done
