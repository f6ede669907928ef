#!/bin/bash
# round 3, run 8: double-pass crossover at mid sizes; config 2 on both HIP runtimes; rocprof evidence for the
# headline (serpentine on and off as separate runs) and the PMC traffic passes
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
for n in 2048 3072 4096 6144; do
  a=$(FWX_DOUBLE_PASS_MIN_N=100000000 python tools/measure_fused.py $n --rates-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  b=$(FWX_DOUBLE_PASS_MIN_N=0 python tools/measure_fused.py $n --rates-only | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  a64=$(FWX_DOUBLE_PASS_MIN_N=100000000 python tools/measure_fused.py $n --rates-only --f64 | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  b64=$(FWX_DOUBLE_PASS_MIN_N=0 python tools/measure_fused.py $n --rates-only --f64 | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['best_ms'])")
  echo "N=$n f32 single $a ms double $b ms | f64 single $a64 ms double $b64 ms" | tee -a $O/r03_double_pass_crossover.txt
done
python tools/measure_perk_small.py > $O/r03_config2_rocm72.json 2>/dev/null; python tools/measure_perk_small.py --torch > $O/r03_config2_torch_rocm70.json 2>/dev/null
python3 -c "
import json
for f in ('r03_config2_rocm72.json','r03_config2_torch_rocm70.json'):
    d=json.load(open('$O/'+f)); print(f, d['hip_runtime'], d['best_ms'], d['solves'][-1]['host_enqueue_ms'], d['solves'][-1]['us_per_launch_by_sixteenth'][:3], d['solves'][-1]['us_per_launch_by_sixteenth'][-2:])
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r03_prof_serp_on -o on -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/r03_bench_rocprof_serp_on.json 2> $O/r03_prof_on.err; echo "serp on rc=$?"
rocprofv3 --kernel-trace --stats -d $O/r03_prof_serp_off -o off -- python3 $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-serpentine > $O/r03_bench_rocprof_serp_off.json 2> $O/r03_prof_off.err; echo "serp off rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r03_pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/r03_bench_under_pmc_fetch.json 2> $O/r03_pmc_f.err; echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r03_pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/r03_bench_under_pmc_write.json 2> $O/r03_pmc_w.err; echo "pmc write rc=$?"
cd $R && python3 tools/pmc_summary.py $O/r03_pmc_f $O/r03_pmc_w $O/r03_pmc_traffic.json | tail -3
ls $O/r03_pmc_f | head -3
