#!/bin/bash
# round 4, run 28: kernel timelines of the other schedules: f64 (+ next-hops) at N = 16384 and 8192, f32 mid sizes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tl() {   # name, then measure_fused arguments
  name=$1; shift
  rm -rf gpurun_out/tl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/measure_fused.py "$@" > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; return 1; }
  f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
  echo "== $name: $*"; grep best_ms gpurun_out/tl.log | cut -c 1-140
  python3 tools/timeline.py "$f" --dump > gpurun_out/r04_timeline_$name.txt; sed -n 1,30p gpurun_out/r04_timeline_$name.txt
  rm -rf gpurun_out/tl
}
tl f64_next_16384 16384 --f64 --next-only &&
tl f64_rates_16384 16384 --f64 --rates-only &&
tl f64_next_8192 8192 --f64 --next-only &&
tl f32_next_4096 4096 --next-only &&
tl f32_next_6144 6144 --next-only &&
tl f32_rates_4096 4096 --rates-only
