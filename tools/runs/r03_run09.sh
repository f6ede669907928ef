#!/bin/bash
# round 3, run 9: multi tests with the enqueue workers; bench rehearsals P = 2 and P = 8 (logical), config 2
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $O/r03_run09_pytest.log 2>&1; rc=$?
tail -3 $O/r03_run09_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --devices 0,0 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_run09_bench_p2.json 2> $O/r03_run09_bench_p2.err; echo "p2 rc=$?"
timeout -k 10 300 python bench.py --devices 0,0,0,0,0,0,0,0 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_run09_bench_p8.json 2> $O/r03_run09_bench_p8.err; echo "p8 rc=$?"
timeout -k 10 300 python bench.py --devices 0,0,0,0,0,0,0,0 --engine fused --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_run09_bench_p8_fused.json 2> $O/r03_run09_bench_p8_fused.err; echo "p8 fused rc=$?"
timeout -k 10 300 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_run09_config2.json 2> $O/r03_run09_config2.err; echo "cfg2 rc=$?"
python3 -c "
import json
for f in ('p2','p8','p8_fused','config2'):
    d=json.loads(open('$O/r03_run09_bench_%s.json'%f if f!='config2' else '$O/r03_run09_config2.json').read())
    print(f, 'ms_per_step', round(d['ms_per_step'],2), 'value %.3e'%d['value'], 'frac', round(d['roofline']['frac'],4), d.get('check'), d.get('exchange'))
"
