#!/bin/bash
# the host segfault of gpu_run42 (fuzz_domain seed 20261005, case 339) once more, under rocgdb, for a native backtrace
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
FUZZ_TRAIL=$O/r02_run43_trail.txt timeout -k 10 330 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop print" -ex run -ex "bt 30" -ex "info threads" -ex "thread apply all bt 12" --args python tools/fuzz_domain.py 230 700 20261005 > $O/r02_run43_gdb.log 2>&1
grep -v "Thread 0x" $O/r02_run43_gdb.log | tail -150 | cut -c1-220
