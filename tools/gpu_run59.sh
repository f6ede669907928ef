#!/bin/bash
# final check of the round: full GPU suite, smoke, mid-size rates timings with the half-width tiles
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run59_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run59_pytest.log; [ $rc -eq 0 ] || { tail -40 $O/r02_run59_pytest.log; exit $rc; }
if grep -l "Memory access fault" $O/r02_run59_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_run59_smoke.log 2>&1 || { tail $O/r02_run59_smoke.log; exit 1; }
tail -1 $O/r02_run59_smoke.log
timeout -k 10 200 python tools/measure_fused.py 2048 4096 6144 8192 --rates-only --check > $O/r02_run59_fused.log 2>&1 || { tail $O/r02_run59_fused.log; exit 1; }
cut -c1-200 $O/r02_run59_fused.log
