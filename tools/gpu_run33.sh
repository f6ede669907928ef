#!/bin/bash
# no look-ahead (one stream, three launches per pass) vs rows-only look-ahead at small orders
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run33.log
for v in 0 100000; do
  echo "FWX_LOOKAHEAD_MIN_N=$v" >> $O/r02_run33.log
  FWX_LOOKAHEAD_MIN_N=$v timeout -k 10 300 python tools/measure_fused.py 256 512 1024 1536 2048 3072 4096 --check >> $O/r02_run33.log 2>&1 || { tail $O/r02_run33.log; exit 1; }
  FWX_LOOKAHEAD_MIN_N=$v timeout -k 10 300 python tools/measure_fused.py 512 1024 2048 3072 --f64 --check >> $O/r02_run33.log 2>&1 || { tail $O/r02_run33.log; exit 1; }
done
python - <<'PY'
import json
cur=None
for l in open('gpurun_out/r02_run33.log'):
    l=l.strip()
    if l.startswith('FWX'): print(l); continue
    if l.startswith('{'):
        d=json.loads(l); print(d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'], d.get('rate_equal_ref'), d.get('next_equal_ref'))
PY
