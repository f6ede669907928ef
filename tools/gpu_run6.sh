#!/bin/bash
# stop at the first failing step; a GPU fault anywhere in the logs fails the run
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
fault() { if grep -l "Memory access fault" $O/r02_run6_*.log 2>/dev/null; then echo "GPU FAULT in the logs above"; exit 9; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_run6_pytest.log 2>&1; rc=$?
tail -8 $O/r02_run6_pytest.log; fault; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/measure_fused.py 16384 --hops --check > $O/r02_run6_fused.log 2>&1 || { tail $O/r02_run6_fused.log; exit 1; }
cut -c1-250 $O/r02_run6_fused.log; fault
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/r02_prof_arg2 -o a --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r02_run6_prof_arg2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $O/r02_sq_f64 -o s --output-format csv -- python3 $R/tools/measure_fused.py 4096 --f64 --rates-only > $O/r02_run6_sq_f64.log 2>&1 || { tail -5 $O/r02_run6_sq_f64.log; exit 1; }
cd $R; fault
python tools/pass_durations.py $O/r02_prof_arg2 fused_main_arg fused_colpanel fused_rowpanel
python - <<'PY'
import csv, glob, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob("gpurun_out/r02_sq_f64/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        n = r["Kernel_Name"]
        k = "main" if "fused_main" in n else "rowpanel" if "rowpanel" in n else "colpanel" if "colpanel" in n else "other"
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[k] += 1
            agg[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, d in agg.items():
    print(k, cnt[k], {c: ("%.4g" % v) for c, v in d.items()})
PY
