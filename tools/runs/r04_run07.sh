#!/bin/bash
# round 4, run 7: ONE WHOLE f64 oracle solve of the headline matrix (N = 16384, before rounding to f32) with
# next-hops, against every GPU engine -> profiles/r04_full_parity_n16384_f64_next.json and the committed
# fixture tests/golden/config4_n16384_f64_digests.json
set -e
cd "$GRAFT_REPO_ROOT"
python tests/golden/make_config4_digests.py gpurun_out/r04_full_parity_n16384_f64_next.json 16384 --next --f64
python tests/golden/make_config4_digests.py --write-fixture-f64 gpurun_out/r04_full_parity_n16384_f64_next.json
cp tests/golden/config4_n16384_f64_digests.json gpurun_out/config4_n16384_f64_digests.json
