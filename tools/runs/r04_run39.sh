#!/bin/bash
# round 4, run 39: the driver's round-end sequence on the final tree (run 15's script), then the host-mirror and per-call
# latencies again (the panel kernels changed)
cd "$GRAFT_REPO_ROOT"
bash tools/runs/r04_run15.sh
python tools/measure_session.py > gpurun_out/r04_session_latency.txt 2>&1; echo "session rc=$?"
python tools/measure_call_latency.py > gpurun_out/r04_call_latency.txt 2>&1; echo "call rc=$?"
tail -4 gpurun_out/r04_session_latency.txt | cut -c 1-330; tail -5 gpurun_out/r04_call_latency.txt | cut -c 1-200
