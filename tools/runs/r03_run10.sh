#!/bin/bash
# round 3, run 10: measurements for DESIGN.md on the round's final kernels
O=gpurun_out
python tools/measure_fused.py 16384 --hops > $O/r03_fused_n16384_f32.jsonl 2>&1; echo "f32 16384 rc=$?"
python tools/measure_fused.py 16384 --f64 --hops > $O/r03_fused_n16384_f64.jsonl 2>&1; echo "f64 16384 rc=$?"
python tools/measure_fused.py 32768 --next-only > $O/r03_fused_n32768_f32.jsonl 2>&1; echo "32768 rc=$?"
python tools/measure_fused.py 8192 4096 1024 > $O/r03_fused_small_f32.jsonl 2>&1; echo "small f32 rc=$?"
python tools/measure_fused.py 8192 4096 1024 --f64 --rates-only > $O/r03_fused_small_f64.jsonl 2>&1; echo "small f64 rc=$?"
for c in 2 3 5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-extras > $O/r03_bench_config$c.json 2> $O/r03_bench_config$c.err; echo "config $c rc=$?"; done
python tools/measure_session.py > $O/r03_session_latency.txt 2>&1; echo "session rc=$?"
python tools/measure_multi.py 16384 > $O/r03_multi_overhead.jsonl 2>&1; echo "multi rc=$?"
cat $O/r03_fused_n16384_f32.jsonl $O/r03_fused_n16384_f64.jsonl $O/r03_fused_n32768_f32.jsonl | cut -c1-200
