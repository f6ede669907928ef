#!/bin/bash
# round 4, run 13: fuzz on the final build -- inside the domain (handles that pad, the pair schedule forced at
# random on one device and on partitions, resumed price changes), hostile values through every engine, and the
# multi-process drivers; default binding (no torch in the fuzzing process)
cd "$GRAFT_REPO_ROOT"
python tools/fuzz_domain.py 240 700 20261005 > gpurun_out/r04_fuzz_domain.txt 2>&1; echo "fuzz_domain rc=$?"; tail -3 gpurun_out/r04_fuzz_domain.txt
python tools/fuzz_long.py 150 200 > gpurun_out/r04_fuzz_long.txt 2>&1; echo "fuzz_long rc=$?"; tail -3 gpurun_out/r04_fuzz_long.txt
python tools/fuzz_dist.py 8 > gpurun_out/r04_fuzz_dist.txt 2>&1; echo "fuzz_dist rc=$?"; tail -3 gpurun_out/r04_fuzz_dist.txt
