#!/bin/bash
# serial vs symmetric at the headline size
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run38.log
for v in 0 100000; do
  echo "FWX_LOOKAHEAD_MIN_N=$v" >> $O/r02_run38.log
  FWX_LOOKAHEAD_MIN_N=$v timeout -k 10 300 python tools/measure_fused.py 16384 --hops >> $O/r02_run38.log 2>&1 || { tail $O/r02_run38.log; exit 1; }
  FWX_LOOKAHEAD_MIN_N=$v timeout -k 10 300 python tools/measure_fused.py 16384 --f64 >> $O/r02_run38.log 2>&1 || { tail $O/r02_run38.log; exit 1; }
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_run38.log'):
    l=l.strip()
    if l.startswith('FWX'): print(l); continue
    if l.startswith('{'):
        d=json.loads(l); print(d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', 'hops' if d['hops'] else '', d['best_ms'])
PY
