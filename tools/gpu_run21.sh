#!/bin/bash
# kernel trace of one f32 rates-only and one +next solve at N=16384 under the symmetric schedule
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_sym -o f --output-format csv -- python3 $R/tools/measure_fused.py 16384 --rates-only > $O/r02_run21_prof.log 2>&1 || { tail $O/r02_run21_prof.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_sym_next -o f --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r02_run21_prof2.log 2>&1 || { tail $O/r02_run21_prof2.log; exit 1; }
cd $R
head -8 $O/r02_prof_sym/f_kernel_stats.csv | cut -c1-200
head -8 $O/r02_prof_sym_next/f_kernel_stats.csv | cut -c1-200
