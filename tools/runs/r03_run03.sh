#!/bin/bash
# round 3, run 3: the domain fuzz ONCE on the default binding and ONCE with torch's bundled HIP runtime
# (after the lifetime hygiene of this round), the one-shot multi call latency, the default bench
O=gpurun_out
python tools/measure_multi_call.py 4096 2 > $O/r03_multi_call.json 2> $O/r03_multi_call.err; echo "multi_call rc=$?"; cat $O/r03_multi_call.json
FUZZ_TRAIL=$O/r03_fuzz_default_trail.txt timeout -k 10 260 python tools/fuzz_domain.py 200 700 20261005 > $O/r03_fuzz_default.log 2>&1; echo "fuzz default rc=$?"; tail -2 $O/r03_fuzz_default.log
FUZZ_IMPORT_TORCH=1 FUZZ_TRAIL=$O/r03_fuzz_torch_trail.txt timeout -k 10 260 python tools/fuzz_domain.py 200 700 20261005 > $O/r03_fuzz_torch.log 2>&1; echo "fuzz torch-runtime rc=$?"; tail -3 $O/r03_fuzz_torch.log
timeout -k 10 400 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "bench rc=$?"; head -c 600 $O/r03_bench_default.json
