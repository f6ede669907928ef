#!/bin/bash
# symmetric look-ahead: forced at small sizes (tests/test_gpu_symmetric.py), then the fused parity tests,
# then timings with a bit check at 8192 and 16384 (default threshold -> symmetric from 8192)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run20_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_symmetric.py -m gpu -q -x --timeout 300 > $O/r02_run20_sym.log 2>&1; rc=$?
tail -5 $O/r02_run20_sym.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_parity.py tests/test_gpu_parity_inputs.py tests/test_gpu_multi.py -m gpu -q -x --timeout 600 > $O/r02_run20_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run20_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/measure_fused.py 8192 16384 --hops --check > $O/r02_run20_a.log 2>&1 || { tail $O/r02_run20_a.log; exit 1; }
fault
FWX_SYMMETRIC_MIN_N=100000 timeout -k 10 300 python tools/measure_fused.py 8192 16384 > $O/r02_run20_off.log 2>&1 || { tail $O/r02_run20_off.log; exit 1; }
FWX_SYMMETRIC_MIN_N=0 timeout -k 10 300 python tools/measure_fused.py 4096 6144 > $O/r02_run20_on_small.log 2>&1 || { tail $O/r02_run20_on_small.log; exit 1; }
timeout -k 10 300 python tools/measure_fused.py 4096 6144 > $O/r02_run20_off_small.log 2>&1 || { tail $O/r02_run20_off_small.log; exit 1; }
echo default; cut -c1-210 $O/r02_run20_a.log; echo off; cut -c1-170 $O/r02_run20_off.log; echo on_small; cut -c1-170 $O/r02_run20_on_small.log; echo off_small; cut -c1-170 $O/r02_run20_off_small.log
