#!/bin/bash
# round 3, run 27: SQ counters of the f64 arg main kernel (N = 16384, rates + next)
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/r03_sq_f64 -o p --output-format csv -- python3 $R/tools/measure_fused.py 16384 --f64 --next-only > $O/r03_run27.log 2>&1 || exit 1
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python3 $R/tools/pmc_by_kernel.py $O/r03_sq_f64 $c | grep "arg_f64" | cut -c1-200; done
