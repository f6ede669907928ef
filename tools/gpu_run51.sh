#!/bin/bash
# the other BASELINE configs through bench.py (2: N=1024 f64, 3: N=8192 f32 fused, 5: N=32768 f32 + next fused)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run51_configs.jsonl
for c in 2 3 5; do
  timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline >> $O/r02_run51_configs.jsonl 2>> $O/r02_run51.err || { tail $O/r02_run51.err; exit 1; }
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_run51_configs.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config'].get('workload','')[:70], '|', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
PY
