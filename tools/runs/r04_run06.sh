#!/bin/bash
# round 4, run 6: fused_main_arg A/B on one box -- wave priority (FWX_EXP_PRIO=2) and row stores after the
# re-scans (FWX_EXP_LATE_STORES=1), each a build variant selected with FWX_LIB_PATH; f32 + next at N=16384,
# f32 + next + trace, config 5 size
set -e
cd "$GRAFT_REPO_ROOT"
for v in base prio late both base; do
  if [ "$v" = base ]; then unset FWX_LIB_PATH; else export FWX_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/libfwx_$v.so; fi
  echo "== $v" 
  python tools/measure_fused.py 16384 --next-only --check 2>&1 | tail -1
  python tools/measure_fused.py 16384 --trace-only 2>&1 | tail -1
done > gpurun_out/r04_arg_ab.txt 2>&1
cat gpurun_out/r04_arg_ab.txt
