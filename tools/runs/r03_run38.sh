#!/bin/bash
# round 3, run 38: same-box A/B: tree instead of chain for the re-scan's maximum (f32 max3 tree, f64 pair tree)
R=$GRAFT_REPO_ROOT
for mode in "" "--f64"; do
  for v in prev new prev new; do [ "$mode" = "" ] && continue
    unset FWX_LIB_PATH
    [ $v = prev ] && export FWX_LIB_PATH=$R/build/libfwx_prev.so
    ms=$(python3 $R/tools/measure_fused.py 16384 $mode --next-only --check | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))")
    echo "N=16384 $mode +next $v: $ms"
  done
done
