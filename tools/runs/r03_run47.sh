#!/bin/bash
# round 3, run 47: f64 rates-only interior path, two pivots per fold call: same-box A/B, then parity
R=$GRAFT_REPO_ROOT
for n in 8192 16384; do
  for v in prev new prev new; do
    unset FWX_LIB_PATH
    [ $v = prev ] && export FWX_LIB_PATH=$R/build/libfwx_prev.so
    ms=$(python3 $R/tools/measure_fused.py $n --f64 --rates-only --check | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['best_ms'], d.get('rate_equal_ref'))")
    echo -n "N=$n f64 rates $v $ms | "
  done; echo
done
unset FWX_LIB_PATH
timeout -k 10 900 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_full_parity.py -m gpu -x -q > gpurun_out/r03_run47_pytest.log 2>&1; rc=$?
tail -2 gpurun_out/r03_run47_pytest.log
exit $rc
