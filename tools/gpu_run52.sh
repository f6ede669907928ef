#!/bin/bash
# final verification of the round: full GPU suite, default bench, kernel-trace stats of the bench, fused table
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run52_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run52_pytest.log 2>&1; rc=$?
tail -5 $O/r02_run52_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_run52_smoke.log 2>&1 || { tail $O/r02_run52_smoke.log; exit 1; }
tail -1 $O/r02_run52_smoke.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > $O/r02_run52_bench.json 2> $O/r02_run52_bench.err || { tail $O/r02_run52_bench.err; exit 1; }
fault
timeout -k 10 500 python tools/measure_fused.py 1024 4096 8192 16384 --hops --check > $O/r02_run52_fused.log 2>&1 || { tail $O/r02_run52_fused.log; exit 1; }
cut -c1-190 $O/r02_run52_fused.log; fault
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_bench6 -o b --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f64-extra > $O/r02_run52_prof_bench.json 2> $O/r02_run52_prof_bench.err || exit 1
cd $R; fault
python -c "
import json
d=json.loads(open('gpurun_out/r02_run52_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['serpentine_off']['frac'], d['f64']['per_k']['roofline']['frac'], d['f64']['fused']['ms_per_step'], d['fused_engine']['ms_per_step'], d['check'], d['cpu_baseline']['value'])
"
head -8 $O/r02_prof_bench6/b_kernel_stats.csv | cut -c1-160
timeout -k 10 300 python tools/measure_fused.py 1024 4096 8192 16384 --f64 --hops > $O/r02_run52_f64.log 2>&1 || { tail $O/r02_run52_f64.log; exit 1; }
cut -c1-170 $O/r02_run52_f64.log
timeout -k 10 200 python tools/measure_fused.py 32768 --next-only > $O/r02_run52_32k.log 2>&1 || { tail $O/r02_run52_32k.log; exit 1; }
cut -c1-170 $O/r02_run52_32k.log
