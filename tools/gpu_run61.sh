#!/bin/bash
# kernel trace of N=16384 f32 + next on the final code: per-pass durations of the arg kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_next_final -o f --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r02_run61.log 2>&1 || { tail $O/r02_run61.log; exit 1; }
cd $R
grep "^{" $O/r02_run61.log | cut -c1-160
python tools/pass_durations.py $O/r02_prof_next_final "fused_main_arg<3, 8>" "fused_main_arg<3, 4>" fused_panels > $O/r02_arg_passes_final.json
cat $O/r02_arg_passes_final.json
