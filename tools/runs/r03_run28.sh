#!/bin/bash
# round 3, run 28: f64 arg kernel with the two column vectors half a tile apart (no W bank conflicts):
# parity suites, timings, LDS counters
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_multi.py tests/test_gpu_full_parity.py tests/test_gpu_resume.py -m gpu -x -q > $O/r03_run28_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run28_pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/measure_fused.py 16384 --f64 --hops 2>&1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/r03_sq_f64b -o p --output-format csv -- python3 $R/tools/measure_fused.py 16384 --f64 --next-only > $O/r03_run28.log 2>&1 || exit 1
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do python3 $R/tools/pmc_by_kernel.py $O/r03_sq_f64b $c | grep "arg_f64" | cut -c1-200; done
