#!/bin/bash
# round 3, run 20: HBM traffic (PMC, separate passes) of the fused main kernels at N = 16384:
# rates + next (arg kernel) and rates only (double pass)
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in next rates; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/r03_pmc_fused_${mode}_$c -o p --output-format csv -- python3 $R/tools/measure_fused.py 16384 --${mode}-only > $O/r03_run20_${mode}_$c.log 2>&1 || exit 1
    python3 $R/tools/pmc_by_kernel.py $O/r03_pmc_fused_${mode}_$c $c | tee -a $O/r03_pmc_fused_by_kernel.jsonl | cut -c1-260
  done
done
