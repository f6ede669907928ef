#!/bin/bash
# EXPERIMENT: 64 x 64 vs 128 x 64 tiles of fused_main_arg after the re-scan rewrite
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r02_run60.log
for v in 1 0; do
  echo "FWX_EXP_ARG_SMALL=$v" >> $O/r02_run60.log
  FWX_EXP_ARG_SMALL=$v timeout -k 10 200 python tools/measure_fused.py 3072 4096 5120 6144 7168 8192 10240 --next-only >> $O/r02_run60.log 2>&1 || exit 1
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_run60.log'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['n'], d['best_ms'])
    else: print(l)
PY
