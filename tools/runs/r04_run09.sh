#!/bin/bash
# round 4, run 9: the round's rocprofv3 evidence for the benchmark command -- kernel-trace stats with the
# serpentine sweep on and off (separate runs), and the two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace
# only) over the timed command, summarised into profiles/r04_pmc_traffic.json
O=$PWD/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r04_prof_on -o p -- python3 $R/bench.py --no-extras --no-cpu-baseline > $O/r04_bench_under_rocprof_on.json 2> $O/r04_prof_on.err || exit 1
python3 $R/tools/rocpd_summary.py $(ls $O/r04_prof_on/*.db $O/r04_prof_on/*/*.db 2>/dev/null | head -1) > $O/r04_bench_kernel_stats_serpentine_on.txt; head -5 $O/r04_bench_kernel_stats_serpentine_on.txt | cut -c1-160
rocprofv3 --kernel-trace --stats -d $O/r04_prof_off -o p -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-serpentine > $O/r04_bench_under_rocprof_off.json 2> $O/r04_prof_off.err || exit 1
python3 $R/tools/rocpd_summary.py $(ls $O/r04_prof_off/*.db $O/r04_prof_off/*/*.db 2>/dev/null | head -1) > $O/r04_bench_kernel_stats_serpentine_off.txt; head -5 $O/r04_bench_kernel_stats_serpentine_off.txt | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r04_pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/r04_bench_under_pmc_fetch.json 2> $O/r04_pmc_f.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r04_pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/r04_bench_under_pmc_write.json 2> $O/r04_pmc_w.err || exit 1
python3 $R/tools/pmc_summary.py $O/r04_pmc_f $O/r04_pmc_w $O/r04_pmc_traffic.json; head -c 900 $O/r04_pmc_traffic.json
# the csv dumps are large: keep only the summaries
rm -rf $O/r04_pmc_f $O/r04_pmc_w $O/r04_prof_on $O/r04_prof_off
