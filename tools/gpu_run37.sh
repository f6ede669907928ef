#!/bin/bash
# crossover between the serial schedule and the symmetric look-ahead, after the merged panels launch
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run37.log
for v in 0 100000; do
  echo "FWX_LOOKAHEAD_MIN_N=$v" >> $O/r02_run37.log
  FWX_LOOKAHEAD_MIN_N=$v timeout -k 10 300 python tools/measure_fused.py 4096 5120 6144 7168 8192 10240 12288 >> $O/r02_run37.log 2>&1 || { tail $O/r02_run37.log; exit 1; }
  FWX_LOOKAHEAD_MIN_N=$v timeout -k 10 300 python tools/measure_fused.py 3072 4096 5120 6144 8192 --f64 >> $O/r02_run37.log 2>&1 || { tail $O/r02_run37.log; exit 1; }
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_run37.log'):
    l=l.strip()
    if l.startswith('FWX'): print(l); continue
    if l.startswith('{'):
        d=json.loads(l); print(d['n'], d['dtype'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', d['best_ms'])
PY
