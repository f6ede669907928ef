#!/bin/bash
# round 4, run 27: main launches that starve the panels go out as two halves (single-device and partitioned double pass):
# schedule tests, then A/B by size and on logical partitions with next-hops
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py tests/test_gpu_multi_double_pass.py tests/test_gpu_part_handle.py -x -q -m gpu > gpurun_out/r04_run27_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r04_run27_tests.log
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do
  echo "== FWX_SPLIT_MAIN=$v"
  FWX_SPLIT_MAIN=$v python tools/measure_fused.py 8192 9216 10240 --next-only --check 2>&1 | cut -c 1-130
  FWX_SPLIT_MAIN=$v python tools/measure_fused.py 8192 --trace-only 2>&1 | cut -c 1-130
done 2>&1 | tee gpurun_out/r04_split_main_ab.txt
python tools/measure_multi.py 16384 --next > gpurun_out/r04_run27_multi_next.json 2> gpurun_out/r04_run27_multi.err; echo "multi rc=$?"; cut -c 1-420 gpurun_out/r04_run27_multi_next.json
