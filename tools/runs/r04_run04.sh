#!/bin/bash
# round 4, run 4: partitioned double pass -- overhead over the single-device handle, rehearsal bench lines
set -e
cd "$GRAFT_REPO_ROOT"
python tools/measure_multi.py 16384 > gpurun_out/r04_multi_overhead_rates.jsonl 2> gpurun_out/r04_multi_overhead_rates.err
python tools/measure_multi.py 16384 --next > gpurun_out/r04_multi_overhead_next.jsonl 2> gpurun_out/r04_multi_overhead_next.err
python bench.py --devices 0,0,0,0,0,0,0,0 --steps 1 --warmup 1 --cpu-seconds 4 > gpurun_out/r04_bench_multi_rehearsal_p8_logical.json 2> gpurun_out/r04_bench_multi_rehearsal_p8.err
python bench.py --devices 0,0 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_multi_rehearsal_p2_logical.json 2> gpurun_out/r04_bench_multi_rehearsal_p2.err
tail -n 3 gpurun_out/r04_multi_overhead_rates.jsonl gpurun_out/r04_multi_overhead_next.jsonl
cut -c 1-1500 gpurun_out/r04_bench_multi_rehearsal_p8_logical.json
