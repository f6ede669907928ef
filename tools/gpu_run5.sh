#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/r02_pytest5.log 2>&1
tail -15 $O/r02_pytest5.log
timeout -k 10 300 python tools/measure_fused.py 16384 --hops --check > $O/r02_hops1.log 2>&1; cut -c1-250 $O/r02_hops1.log
timeout -k 10 200 python tools/measure_fused.py 32768 --next-only > $O/r02_cfg5.log 2>&1; cut -c1-250 $O/r02_cfg5.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/r02_clk_f64 -o c --output-format csv -- python3 $R/tools/measure_fused.py 16384 --f64 --rates-only > $O/r02_clk_f64.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/r02_clk_f32 -o c --output-format csv -- python3 $R/tools/measure_fused.py 16384 --next-only > $O/r02_clk_f32.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections
for tag in ("r02_clk_f64", "r02_clk_f32"):
    agg = collections.defaultdict(lambda: [0.0, 0, 0])
    for path in glob.glob(os.path.join("gpurun_out", tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            k = r["Kernel_Name"].split("(")[0][-60:]
            a = agg[k]
            a[0] += float(r["Counter_Value"]); a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[2] += 1
    for k, (cyc, ns, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:4]:
        print(tag, k, "launches", cnt, "avg_us %.1f" % (ns / cnt / 1e3), "GHz %.3f" % (cyc / ns))
PY
