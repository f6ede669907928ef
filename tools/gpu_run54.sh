#!/bin/bash
# fuzz after the pinned staging of small host transfers (system HIP runtime), then the full GPU suite once more
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
FWX_NO_TORCH=1 FUZZ_TRAIL=$O/r02_run54_trail1.txt timeout -k 10 150 python tools/fuzz_domain.py 100 400 20261010 > $O/r02_run54_a.log 2>&1; rc=$?
tail -1 $O/r02_run54_a.log | cut -c1-200; [ $rc -eq 0 ] || { tail -20 $O/r02_run54_a.log; cat $O/r02_run54_trail1.txt; exit $rc; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run54_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run54_pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_run54_smoke.log 2>&1 || { tail $O/r02_run54_smoke.log; exit 1; }
tail -1 $O/r02_run54_smoke.log
