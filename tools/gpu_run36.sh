#!/bin/bash
# write-back without read-modify-write in the compare-form kernel, diagonal restore hoisted everywhere:
# all fused parity tests, then f64 (+next = compare form) and f32 timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run36_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_full_parity.py tests/test_gpu_parity_inputs.py tests/test_gpu_multi.py -m gpu -q -x --timeout 600 > $O/r02_run36_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run36_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/measure_fused.py 1024 4096 8192 --f64 --check > $O/r02_run36_f64.log 2>&1 || { tail $O/r02_run36_f64.log; exit 1; }
fault
timeout -k 10 300 python tools/measure_fused.py 256 512 1024 2048 4096 6144 8192 --check > $O/r02_run36_f32.log 2>&1 || { tail $O/r02_run36_f32.log; exit 1; }
cut -c1-220 $O/r02_run36_f64.log $O/r02_run36_f32.log
