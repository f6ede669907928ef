#!/bin/bash
# round 3, run 24: A/B on one box, old (build/libfwx_old.so = previous commit) against new arg kernels, with
# the path trace: wall time, then standalone kernel durations (a PMC pass serialises the launches)
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
for v in old new; do
  [ $v = old ] && export FWX_LIB_PATH=$R/build/libfwx_old.so || unset FWX_LIB_PATH
  echo "== $v"; python3 $R/tools/measure_fused.py 16384 --trace-only | cut -c1-150
  python3 $R/tools/measure_fused.py 16384 --next-only | cut -c1-150
done
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  [ $v = old ] && export FWX_LIB_PATH=$R/build/libfwx_old.so || unset FWX_LIB_PATH
  rocprofv3 --kernel-trace --pmc SQ_WAVES -d $O/r03_ab_$v -o p --output-format csv -- python3 $R/tools/measure_fused.py 16384 --trace-only > $O/r03_run24_$v.log 2>&1 || exit 1
  echo "== $v standalone"; python3 $R/tools/pmc_by_kernel.py $O/r03_ab_$v SQ_WAVES | head -4 | cut -c1-220
done
