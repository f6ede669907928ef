#!/bin/bash
# round 4, run 14: fused_main_arg_f64 with 32 resident pivots per sub-pass (FWX_ARG_F64_HALF_STRIPS=1: three
# workgroups per CU) against the shipped 64 (two per CU): correctness under the switch, then A/B on one box
cd "$GRAFT_REPO_ROOT"
FWX_ARG_F64_HALF_STRIPS=1 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py -x -q -m gpu -k "float64 or f64 or fuzz or exact" > gpurun_out/r04_run14_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_run14_tests.log
for v in 0 1 0 1; do
  export FWX_ARG_F64_HALF_STRIPS=$v
  echo "== half strips = $v"
  python tools/measure_fused.py 16384 --f64 --next-only --check 2>&1 | tail -1 | cut -c 1-230
  python tools/measure_fused.py 16384 --f64 --trace-only 2>&1 | tail -1 | cut -c 1-200
done > gpurun_out/r04_f64_half_ab.txt 2>&1
cat gpurun_out/r04_f64_half_ab.txt
