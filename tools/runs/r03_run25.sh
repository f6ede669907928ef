#!/bin/bash
# round 3, run 25: SQ counters of the arg main kernel (N = 16384, rates + next), two passes
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $O/r03_sq_a -o p --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r03_run25_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA -d $O/r03_sq_b -o p --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r03_run25_b.log 2>&1 || exit 1
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES; do python3 $R/tools/pmc_by_kernel.py $O/r03_sq_a $c | grep "arg<3, 8>" | cut -c1-200; done
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA; do python3 $R/tools/pmc_by_kernel.py $O/r03_sq_b $c | grep "arg<3, 8>" | cut -c1-200; done
