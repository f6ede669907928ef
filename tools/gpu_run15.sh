#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for th in 512 1100 2400 4200; do
  echo "== small-tile threshold $th"
  FWX_SMALL_TILES=$th timeout -k 10 200 python tools/measure_fused.py 2048 3072 4096 6144 8192 --rates-only || exit 1
  FWX_SMALL_TILES=$th timeout -k 10 200 python tools/measure_fused.py 4096 6144 --next-only || exit 1
done > $O/r02_run15_tiles.log 2>&1
python - <<'PY'
import json
th=None
for l in open('gpurun_out/r02_run15_tiles.log'):
    if l.startswith('=='): th=l.strip()
    elif l.startswith('{'):
        d=json.loads(l); print(th, d['n'], 'next' if d['next'] else 'rates', d['best_ms'])
PY
