#!/bin/bash
# fused_main_arg_f64: all fused parity tests (f64 + next / hops / trace now take it), then timings with a bit check
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run48_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_full_parity.py tests/test_gpu_parity_inputs.py tests/test_gpu_multi.py tests/test_gpu_host_session.py -m gpu -q -x --timeout 600 > $O/r02_run48_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run48_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/measure_fused.py 1024 4096 8192 16384 --f64 --hops --check > $O/r02_run48_f64.log 2>&1 || { tail $O/r02_run48_f64.log; exit 1; }
fault
cut -c1-220 $O/r02_run48_f64.log
