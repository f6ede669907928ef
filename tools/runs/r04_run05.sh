#!/bin/bash
# round 4, run 5: resume on partitioned handles, config-5 pinning, f64 whole-oracle parity
set -e
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_resume.py tests/test_gpu_host_session.py -x -q -m gpu > gpurun_out/r04_run05_a.log 2>&1 || { tail -30 gpurun_out/r04_run05_a.log; exit 1; }
tail -3 gpurun_out/r04_run05_a.log
python -m pytest tests/test_gpu_full_parity.py -x -q -m gpu --durations=5 > gpurun_out/r04_run05_b.log 2>&1 || { tail -30 gpurun_out/r04_run05_b.log; exit 1; }
tail -9 gpurun_out/r04_run05_b.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "config5" --durations=5 > gpurun_out/r04_run05_c.log 2>&1 || { tail -40 gpurun_out/r04_run05_c.log; exit 1; }
tail -9 gpurun_out/r04_run05_c.log
