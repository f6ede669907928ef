#!/bin/bash
# round 4, run 21: EXPERIMENT -- side stream at the highest stream priority / compute units reserved for the side chain
# (main stream under a CU mask), at the sizes where the side chain bounds the double pass
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $1"
  python tools/measure_fused.py 4096 6144 8192 12288 --next-only --check 2>&1 | cut -c 1-150
  python tools/measure_fused.py 4096 6144 8192 12288 --rates-only --check 2>&1 | cut -c 1-150
}
{
run base
FWX_SIDE_PRIORITY=1 run priority
FWX_MAIN_CU_RESERVE=8 run reserve8
FWX_MAIN_CU_RESERVE=16 run reserve16
FWX_MAIN_CU_RESERVE=32 run reserve32
FWX_SIDE_PRIORITY=1 FWX_MAIN_CU_RESERVE=16 run priority+reserve16
run base
} > gpurun_out/r04_side_chain_streams.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04_side_chain_streams.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    try:
        i=l.index('"best_ms"'); n=l[l.index('"n"')+5:].split(',')[0]; nx='"next": true' in l
        print(' ', n, 'next' if nx else 'rates', l[i:i+20], 'ok' if '"rate_equal_ref": true' in l else '??')
    except Exception: print(l[:150])
PY
