#!/bin/bash
# full suite + default bench + config benches (stop at the first failure; a GPU fault fails the run)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run9_*.log $O/r02_run9_*.err 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run9_pytest.log 2>&1; rc=$?
tail -6 $O/r02_run9_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/r02_run9_bench.json 2> $O/r02_run9_bench.err || { tail $O/r02_run9_bench.err; exit 1; }
fault
for c in 2 3 5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/r02_run9_cfg$c.json 2> $O/r02_run9_cfg$c.err || { tail $O/r02_run9_cfg$c.err; exit 1; }; done
fault
python -c "
import json
for f in ('r02_run9_bench','r02_run9_cfg2','r02_run9_cfg3','r02_run9_cfg5'):
    d=json.loads(open('gpurun_out/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), json.dumps(d.get('f64',{}))[:700], json.dumps(d.get('fused_engine',{}))[:500], d.get('check'))
"
