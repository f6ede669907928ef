#!/bin/bash
# round 4, run 33: fuzz on the build with the side chain's wave priorities and the split main launches (second use: after the panel flags, seed 20261007)
cd "$GRAFT_REPO_ROOT"
python tools/fuzz_domain.py 200 900 20261007 > gpurun_out/r04_fuzz_domain_b.txt 2>&1; echo "fuzz_domain rc=$?"; tail -3 gpurun_out/r04_fuzz_domain_b.txt
python tools/fuzz_long.py 100 200 > gpurun_out/r04_fuzz_long_b.txt 2>&1; echo "fuzz_long rc=$?"; tail -3 gpurun_out/r04_fuzz_long_b.txt
