#!/bin/bash
# round 3, run 45: 64 x 64 arg kernel, two pivot pairs per trip: same-box A/B at mid sizes, then parity
R=$GRAFT_REPO_ROOT
for n in 1024 2048 4096 6144 8192; do
  for v in prev new prev new; do
    unset FWX_LIB_PATH
    [ $v = prev ] && export FWX_LIB_PATH=$R/build/libfwx_prev.so
    ms=$(python3 $R/tools/measure_fused.py $n --next-only | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['best_ms'])")
    echo -n "N=$n $v $ms | "
  done; echo
done
unset FWX_LIB_PATH
timeout -k 10 900 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r03_run45_pytest.log 2>&1; rc=$?
tail -2 gpurun_out/r03_run45_pytest.log
exit $rc
