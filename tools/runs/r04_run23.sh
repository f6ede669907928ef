#!/bin/bash
# round 4, run 23: the side chain's kernels at raised wave priority (panels 3, look-ahead launches 2) -- schedule tests,
# then solve times by size and the logical-partition overheads
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py tests/test_gpu_multi_double_pass.py -x -q -m gpu > gpurun_out/r04_run23_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r04_run23_tests.log
[ $rc -eq 0 ] || exit $rc
python tools/measure_fused.py 1024 2048 4096 6144 8192 12288 16384 --rates-only --check 2>&1 | cut -c 1-120 | tee gpurun_out/r04_run23_rates.txt &&
python tools/measure_fused.py 1024 2048 4096 6144 8192 12288 16384 --next-only --check 2>&1 | cut -c 1-120 | tee gpurun_out/r04_run23_next.txt &&
python tools/measure_multi.py > gpurun_out/r04_run23_multi.json 2> gpurun_out/r04_run23_multi.err; echo "multi rc=$?"; tail -c 1500 gpurun_out/r04_run23_multi.json
