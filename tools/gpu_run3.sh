#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in 0 1 2 3; do echo "variant $v"; FWX_MAXF_VARIANT=$v timeout -k 10 120 python tools/measure_fused.py 16384 --rates-only || exit 1; done > $O/r02_maxf_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r02_prof_arg -o a --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r02_prof_arg.log 2>&1
cd $R
python tools/pass_durations.py $O/r02_prof_arg fused_main_arg fused_colpanel fused_rowpanel > $O/r02_arg_passes.json
cat $O/r02_maxf_ab.log | grep -E "variant|best_ms" | cut -c1-200
cat $O/r02_arg_passes.json
