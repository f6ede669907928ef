#!/bin/bash
# round 3, run 33: same-box A/B: previous build (single pass only) against the looped arg kernels, single and double pass
R=$GRAFT_REPO_ROOT
for mode in "" "--f64"; do
  for v in prev new-single new-double prev new-single new-double; do
    unset FWX_LIB_PATH FWX_DOUBLE_PASS_NEXT_MIN_N
    [ $v = prev ] && export FWX_LIB_PATH=$R/build/libfwx_prev.so
    [ $v = new-single ] && export FWX_DOUBLE_PASS_NEXT_MIN_N=100000000
    ms=$(python3 $R/tools/measure_fused.py 16384 $mode --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms'])")
    echo "N=16384 $mode +next $v: $ms"
  done
done
