#!/bin/bash
# round 3, run 30: domain fuzz and the long hostile fuzz once on the final arg kernels (assembly
# compaction, four-wide tracking, split f64 columns), default binding; then with the 32 x 64 f64 tiles
O=gpurun_out
FUZZ_TRAIL=$O/r03_fuzz_final_trail.txt timeout -k 10 260 python tools/fuzz_domain.py 200 900 20261021 > $O/r03_fuzz_final.log 2>&1; rc=$?; echo "fuzz_domain rc=$rc"; tail -1 $O/r03_fuzz_final.log
[ $rc -ne 0 ] && exit $rc
FWX_ARG_F64_SHORT_TILES=1 FWX_ARG_SMALL_TILES_BELOW=0 timeout -k 10 160 python tools/fuzz_domain.py 100 900 20261022 > $O/r03_fuzz_final_b.log 2>&1; rc=$?; echo "fuzz_domain (short f64 tiles, 128x64 f32 tiles) rc=$rc"; tail -1 $O/r03_fuzz_final_b.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/fuzz_long.py 150 > $O/r03_fuzz_long_final.log 2>&1; rc=$?; echo "fuzz_long rc=$rc"; tail -1 $O/r03_fuzz_long_final.log
exit $rc
