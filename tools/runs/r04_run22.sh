#!/bin/bash
# round 4, run 22: EXPERIMENT -- wave priority (s_setprio) for the side chain's kernels: panels at 3 (variant pprio),
# panels at 3 + cross launches at 2 (variant cprio), against the shipped build
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $1"
  python tools/measure_fused.py 2048 4096 6144 8192 12288 16384 --next-only --check 2>&1 | cut -c 1-240
  python tools/measure_fused.py 2048 4096 6144 8192 12288 16384 --rates-only --check 2>&1 | cut -c 1-240
}
{
run base
FWX_LIB_PATH=$PWD/build/variants/libfwx_pprio.so run panels3
FWX_LIB_PATH=$PWD/build/variants/libfwx_cprio.so run panels3+cross2
run base
FWX_LIB_PATH=$PWD/build/variants/libfwx_pprio.so run panels3
FWX_LIB_PATH=$PWD/build/variants/libfwx_cprio.so run panels3+cross2
} > gpurun_out/r04_side_chain_setprio.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04_side_chain_setprio.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    try:
        i=l.index('"best_ms"'); n=l[l.index('"n"')+5:].split(',')[0]; nx='"next": true' in l
        print(' ', n, 'next' if nx else 'rates', l[i:i+20], 'ok' if '"rate_equal_ref": true' in l else '??')
    except Exception: print(l[:150])
PY
