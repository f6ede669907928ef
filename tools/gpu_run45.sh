#!/bin/bash
# the same fuzz with the SYSTEM HIP runtime (FWX_NO_TORCH=1: /opt/rocm, ROCm 7.2) instead of the one bundled with torch
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
export FWX_NO_TORCH=1
FWX_LOOKAHEAD_MIN_N=0 FWX_SYMMETRIC_MIN_N=0 FUZZ_TRAIL=$O/r02_run45_trail1.txt timeout -k 10 260 python tools/fuzz_domain.py 200 600 20261006 > $O/r02_run45_fuzz_sym.log 2>&1; rc=$?
tail -2 $O/r02_run45_fuzz_sym.log | cut -c1-200; cat $O/r02_run45_trail1.txt
python - <<'PY'
import sys
sys.path.insert(0,'.')
PY
grep -c "libamdhip64" /proc/self/maps > /dev/null 2>&1
exit $rc
