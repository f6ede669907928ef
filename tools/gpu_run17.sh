#!/bin/bash
# A/B of the explicit "tile complete" wait in fused_main_max / fused_main_max_f64: rates-only timings with a
# bit check against the per-k engine at 8192, then 16384 f32 and f64.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run17_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi; }
timeout -k 10 300 python tools/measure_fused.py 8192 --rates-only --check > $O/r02_run17_a.log 2>&1 || { tail $O/r02_run17_a.log; exit 1; }
fault
timeout -k 10 300 python tools/measure_fused.py 4096 16384 --rates-only >> $O/r02_run17_a.log 2>&1 || { tail $O/r02_run17_a.log; exit 1; }
timeout -k 10 300 python tools/measure_fused.py 8192 --f64 --rates-only --check > $O/r02_run17_b.log 2>&1 || { tail $O/r02_run17_b.log; exit 1; }
fault
timeout -k 10 300 python tools/measure_fused.py 16384 --f64 --rates-only >> $O/r02_run17_b.log 2>&1 || { tail $O/r02_run17_b.log; exit 1; }
cut -c1-200 $O/r02_run17_a.log $O/r02_run17_b.log
