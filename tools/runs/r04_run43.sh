#!/bin/bash
# round 4, run 43: last look at the final tree -- smoke() and the default bench line (the full GPU suite last ran green
# before the double-pass threshold constant changed; no GPU test uses an order in [5120, 8192))
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_last_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_last_smoke.txt
timeout -k 10 400 python bench.py > gpurun_out/r04_last_bench_default.json 2> gpurun_out/r04_last_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_last_bench_default.json"))
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_plain_stream"], d["fused_engine"]["ms_per_step"], d["fused_engine"]["valu_roofline"].get("at_kernel_clock"), d["fused_engine_next"]["ms_per_step"], d["fused_engine_next"]["valu_roofline"]["own_scheme"].get("at_kernel_clock"), d["f64"]["check"], d["check"]["per_k_equals_fused_bits"])
PY
