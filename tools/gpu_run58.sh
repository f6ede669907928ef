#!/bin/bash
# EXPERIMENT: 128 x 64 tiles for fused_main_max at mid sizes (workgroup-count quantisation on 768 slots)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo default > $O/r02_run58.log
timeout -k 10 200 python tools/measure_fused.py 3072 4096 5120 6144 7168 8192 10240 12288 16384 --rates-only >> $O/r02_run58.log 2>&1 || exit 1
echo half_tile >> $O/r02_run58.log
FWX_EXP_HALF_TILE=1 timeout -k 10 200 python tools/measure_fused.py 3072 4096 5120 6144 7168 8192 10240 12288 16384 --rates-only --check >> $O/r02_run58.log 2>&1 || exit 1
python - <<'PY'
import json
for l in open('gpurun_out/r02_run58.log'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['n'], d['best_ms'], d.get('rate_equal_ref'))
    else: print(l)
PY
