#!/bin/bash
# interior fast paths (f32 + f64 max kernels): fused-engine parity tests, timings, kernel-trace stats of one f32 solve
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run19_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_parity.py tests/test_gpu_parity_inputs.py tests/test_gpu_multi.py -m gpu -q -x --timeout 600 > $O/r02_run19_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run19_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_fused.py 8192 --f64 --rates-only --check > $O/r02_run19_b.log 2>&1 || { tail $O/r02_run19_b.log; exit 1; }
fault
timeout -k 10 300 python tools/measure_fused.py 4096 16384 --f64 --rates-only >> $O/r02_run19_b.log 2>&1 || { tail $O/r02_run19_b.log; exit 1; }
cut -c1-200 $O/r02_run19_b.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_fast -o f --output-format csv -- python3 $R/tools/measure_fused.py 16384 --rates-only > $O/r02_run19_prof.log 2>&1 || { tail $O/r02_run19_prof.log; exit 1; }
cd $R
head -8 $O/r02_prof_fast/f_kernel_stats.csv | cut -c1-200
python tools/pass_durations.py $O/r02_prof_fast fused_main_max fused_rowpanel fused_colpanel
