#!/bin/bash
# round 3, run 42: fused_panels held to 48 VGPRs (f32 + next, with / without trace): does the side chain now run
# beside the main launch?  A/B timings, then a kernel trace, then parity
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
for mode in "--next-only" "--trace-only"; do
  for v in 0 1 0 1; do
    ms=$(FWX_PANELS_TIGHT=$v python3 tools/measure_fused.py 16384 $mode | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms'])")
    echo "N=16384 f32 $mode tight=$v: $ms"
  done
done
for n in 4096 8192; do for v in 0 1; do
  ms=$(FWX_PANELS_TIGHT=$v python3 tools/measure_fused.py $n --next-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms'])")
  echo "N=$n f32 next tight=$v: $ms"
done; done
python3 tools/measure_fused.py 32768 --next-only | cut -c1-160
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/r03_final_prof_next_tight -o p -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r03_run42_prof.log 2>&1 && python3 $R/tools/rocpd_summary.py $(ls $O/r03_final_prof_next_tight/*.db | head -1) 2>/dev/null | sed -n 1,4p | cut -c1-150)
timeout -k 10 900 python -m pytest tests/test_gpu_double_pass.py tests/test_gpu_symmetric.py tests/test_gpu_parity.py tests/test_gpu_resume.py -m gpu -x -q > $O/r03_run42_pytest.log 2>&1; rc=$?
tail -2 $O/r03_run42_pytest.log
exit $rc
