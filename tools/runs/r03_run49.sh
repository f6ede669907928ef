#!/bin/bash
# round 3, run 49: the driver's round-end sequence on the final tree: pytest -m gpu, smoke(), default bench
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r03_run49_pytest.log 2>&1; rc=$?
tail -4 $O/r03_run49_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; rc=$?
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $O/r03_run49_bench_default.json 2> $O/r03_run49_bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/r03_run49_bench_default.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['fused_engine']['ms_per_step'], d['fused_engine']['valu_roofline']['at_measured_stream_rate']['frac'], d['f64']['fused']['ms_per_step'], d['check'])
"
