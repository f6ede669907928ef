# rocprofv3 kernel-trace summary of the default bench command (profiles/r01_bench_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench/bench_under_rocprof.json 2> $R/gpurun_out/prof_bench/err.log
ls $R/gpurun_out/prof_bench
tail -c 400 $R/gpurun_out/prof_bench/bench_under_rocprof.json
