#!/bin/bash
# EXPERIMENT: the look-ahead (side) stream at the highest stream priority
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo default > $O/r02_run62.log
timeout -k 10 200 python tools/measure_fused.py 16384 --hops >> $O/r02_run62.log 2>&1 || exit 1
echo side_prio >> $O/r02_run62.log
FWX_EXP_SIDE_PRIO=1 timeout -k 10 200 python tools/measure_fused.py 16384 --hops >> $O/r02_run62.log 2>&1 || exit 1
python - <<'PY'
import json
for l in open('gpurun_out/r02_run62.log'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['n'], 'next' if d['next'] else 'rates', 'trace' if d['trace'] else '', 'hops' if d['hops'] else '', d['best_ms'])
    else: print(l)
PY
