#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 200 python tools/measure_dist_single.py 4096 > $O/r02_run11_dist1.log 2>&1 || { tail $O/r02_run11_dist1.log; exit 1; }
tail -1 $O/r02_run11_dist1.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "caller_supplied or overlap" > $O/r02_run11_pytest.log 2>&1; rc=$?
tail -4 $O/r02_run11_pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/measure_session.py > $O/r02_run11_session.log 2>&1 || { tail $O/r02_run11_session.log; exit 1; }
cat $O/r02_run11_session.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r02_prof_session -o s --output-format csv -- python3 $R/tools/measure_session.py > $O/r02_run11_prof_session.log 2>&1 || exit 1
head -12 $O/r02_prof_session/s_kernel_stats.csv | cut -c1-200
if grep -l "Memory access fault" $O/r02_run11_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
