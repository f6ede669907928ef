#!/bin/bash
# round 4, run 24: kernel timeline of the N = 8192 solves (config 3): main launches against the side chain
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for mode in next rates; do
  rm -rf gpurun_out/tl_$mode
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$mode -- python3 tools/measure_fused.py 8192 --$mode-only > gpurun_out/tl_$mode.log 2>&1 || { tail -5 gpurun_out/tl_$mode.log; exit 1; }
  f=$(find gpurun_out/tl_$mode -name '*kernel_trace.csv' | head -1)
  echo "== $mode ($f)"; tail -1 gpurun_out/tl_$mode.log | cut -c 1-160
  python3 tools/timeline.py "$f" --dump | tee gpurun_out/r04_timeline_8192_$mode.txt
  rm -rf gpurun_out/tl_$mode
done
