#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run12_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run12_pytest.log 2>&1; rc=$?
tail -6 $O/r02_run12_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_fused.py 1024 4096 8192 16384 --check > $O/r02_run12_fused.log 2>&1 || { tail $O/r02_run12_fused.log; exit 1; }
cut -c1-200 $O/r02_run12_fused.log; fault
timeout -k 10 200 python tools/measure_fused.py 1024 4096 --f64 > $O/r02_run12_fused64.log 2>&1 || exit 1
cut -c1-200 $O/r02_run12_fused64.log
timeout -k 10 200 python tools/measure_session.py > $O/r02_run12_session.log 2>&1 || exit 1
grep vertices $O/r02_run12_session.log; fault
