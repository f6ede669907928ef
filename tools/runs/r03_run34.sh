#!/bin/bash
# round 3, run 34: why is the looped f64 arg kernel slower?  A = tile load after the staging, B = no opaque thread index
R=$GRAFT_REPO_ROOT
export FWX_DOUBLE_PASS_NEXT_MIN_N=100000000
for v in prev new expA expB expAB; do
  unset FWX_LIB_PATH
  [ $v = prev ] && export FWX_LIB_PATH=$R/build/libfwx_prev.so
  [ ${v#exp} != $v ] && export FWX_LIB_PATH=$R/build/libfwx_$v.so
  ms=$(python3 $R/tools/measure_fused.py 16384 --f64 --next-only --check | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))")
  echo "N=16384 f64 +next single pass, $v: $ms"
done
