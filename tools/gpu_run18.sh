#!/bin/bash
# interior fast path of fused_main_max: parity tests that exercise the fused engine, then timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run18_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_parity.py tests/test_gpu_parity_inputs.py tests/test_gpu_multi.py -m gpu -q -x --timeout 600 > $O/r02_run18_pytest.log 2>&1; rc=$?
tail -5 $O/r02_run18_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_fused.py 8192 --rates-only --check > $O/r02_run18_a.log 2>&1 || { tail $O/r02_run18_a.log; exit 1; }
fault
timeout -k 10 300 python tools/measure_fused.py 2048 4096 6144 16384 --rates-only >> $O/r02_run18_a.log 2>&1 || { tail $O/r02_run18_a.log; exit 1; }
cut -c1-200 $O/r02_run18_a.log
