#!/bin/bash
# round-2 GPU call 1: tests, default bench, kernel-trace stats, PMC passes over the bench command
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
ok_unless_killed() { "$@"; rc=$?; echo "rc=$rc: $*" >> $O/r02_run1_status.txt; [ $rc -ne 124 ] && [ $rc -ne 137 ]; }
cd $R
ok_unless_killed timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_pytest1.log 2>&1 &&
ok_unless_killed timeout -k 10 400 python bench.py > $O/r02_bench1.json 2> $O/r02_bench1.err &&
cd /tmp && export TMPDIR=/tmp &&
ok_unless_killed timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_bench -o b --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f64-extra > $O/r02_prof_bench.json 2> $O/r02_prof_bench.err &&
ok_unless_killed timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/r02_pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/r02_pmc_f.json 2> $O/r02_pmc_f.err &&
ok_unless_killed timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/r02_pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $O/r02_pmc_w.json 2> $O/r02_pmc_w.err
cat $O/r02_run1_status.txt
tail -5 $O/r02_pytest1.log
