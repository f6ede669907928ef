#!/bin/bash
# pooled per-call contexts: the whole GPU suite (host API, threads, device API), then call latency and small sizes
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run40_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run40_pytest.log 2>&1; rc=$?
tail -3 $O/r02_run40_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_call_latency.py > $O/r02_call_latency_after.txt 2>&1 || { tail $O/r02_call_latency_after.txt; exit 1; }
cat $O/r02_call_latency_after.txt
timeout -k 10 300 python tools/measure_small.py > $O/r02_measure_small_after.txt 2>&1 || { tail $O/r02_measure_small_after.txt; exit 1; }
cat $O/r02_measure_small_after.txt
timeout -k 10 300 python tools/measure_session.py > $O/r02_run40_session.txt 2>&1 || { tail $O/r02_run40_session.txt; exit 1; }
cat $O/r02_run40_session.txt
