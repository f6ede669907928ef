#!/bin/bash
# round 4, run 38: with the panel kernels at 32 registers (f32 + next-hops, no trace) a panel workgroup fits when ONE
# workgroup of the 64 x 64 fused_main_arg retires: is the split main launch still needed?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for v in 1 0 1 0; do
  echo "== FWX_SPLIT_MAIN=$v"
  FWX_SPLIT_MAIN=$v python tools/measure_fused.py 8192 9216 10240 --next-only 2>&1 | cut -c 1-120
  FWX_SPLIT_MAIN=$v python tools/measure_fused.py 8192 --trace-only 2>&1 | cut -c 1-120
done 2>&1 | tee gpurun_out/r04_split_main_ab2.txt
rm -rf gpurun_out/tl
FWX_SPLIT_MAIN=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/measure_fused.py 8192 --next-only > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$f" --dump | head -32 | tee gpurun_out/r04_timeline_8192_next_nosplit_panels32.txt
rm -rf gpurun_out/tl
