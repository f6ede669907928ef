#!/bin/bash
# max-form panels (rates only, inside the domain): parity tests, then small / mid / large rates-only timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run50_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_full_parity.py tests/test_gpu_parity_inputs.py -m gpu -q -x --timeout 600 > $O/r02_run50_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run50_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_fused.py 256 512 1024 2048 4096 8192 16384 --rates-only --check > $O/r02_run50_f32.log 2>&1 || { tail $O/r02_run50_f32.log; exit 1; }
timeout -k 10 300 python tools/measure_fused.py 512 1024 4096 --f64 --rates-only --check > $O/r02_run50_f64.log 2>&1 || { tail $O/r02_run50_f64.log; exit 1; }
cut -c1-200 $O/r02_run50_f32.log $O/r02_run50_f64.log
