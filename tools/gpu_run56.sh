#!/bin/bash
# the default bench line and its kernel-trace stats on the final code of the round
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/r02_run56_bench.json 2> $O/r02_run56_bench.err || { tail $O/r02_run56_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_bench7 -o b --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f64-extra > $O/r02_run56_prof_bench.json 2> $O/r02_run56_prof_bench.err || exit 1
cd $R
python -c "
import json
d=json.loads(open('gpurun_out/r02_run56_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['fused_engine']['ms_per_step'], d['f64']['fused']['ms_per_step'], d['check']['per_k_equals_fused_bits'], d['cpu_baseline']['value'])
print(d['reference_regime']['rows'])
"
head -4 $O/r02_prof_bench7/b_kernel_stats.csv | cut -c1-150
