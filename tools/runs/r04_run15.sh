#!/bin/bash
# round 4, run 15: the driver's round-end sequence on the final tree -- pytest -m gpu (timed), smoke(), the default
# bench -- then the config presets and the N > 1 rehearsal lines (logical partitions of the one GPU)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_final
mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 ) > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -16 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for c in 2 3 5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err; echo "config $c rc=$?"; done
timeout -k 10 300 python bench.py --devices 0,0,0,0,0,0,0,0 --steps 1 --warmup 1 --cpu-seconds 3 > $O/bench_multi_rehearsal_p8_logical.json 2> $O/bench_multi_p8.err; echo "p8 rc=$?"
timeout -k 10 300 python bench.py --devices 0,0 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_multi_rehearsal_p2_logical.json 2> $O/bench_multi_p2.err; echo "p2 rc=$?"
python - <<'PY'
import json
O="gpurun_out/r04_final/"
d=json.load(open(O+"bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_plain_stream"], d["per_k_with_next"]["frac"], d["fused_engine"]["ms_per_step"], d["fused_engine_next"]["ms_per_step"], d["fused_engine_next"]["valu_roofline"]["own_scheme"]["frac"], d["f64"]["check"], d["check"]["per_k_equals_fused_bits"])
for c in (2,3,5):
    x=json.load(open(O+"bench_config%d.json"%c)); print("config",c,x["ms_per_step"],x["value"])
for p in (8,2):
    x=json.load(open(O+"bench_multi_rehearsal_p%d_logical.json"%p)); print("P",p,x["ms_per_step"],x["fused_engine"]["ms_per_step"],x["exchange"]["chain_over_bulk"],x["fused_engine"]["exchange"]["chain_over_bulk"],x["check"])
PY
