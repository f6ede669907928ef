#!/bin/bash
# round 4, run 18: host-mirror latencies on the round's build -- plain handle (pads odd orders itself now) and the
# resident matrix over 4 logical partitions (resumes since this round); per-call latency of the one-shot entry points
cd "$GRAFT_REPO_ROOT"
python tools/measure_session.py > gpurun_out/r04_session_latency.txt 2>&1; echo "rc=$?"
python tools/measure_session.py --devices 0,0,0,0 > gpurun_out/r04_session_latency_p4.txt 2>&1; echo "rc=$?"
python tools/measure_call_latency.py > gpurun_out/r04_call_latency.txt 2>&1; echo "rc=$?"
tail -4 gpurun_out/r04_session_latency.txt | cut -c 1-330; tail -3 gpurun_out/r04_session_latency_p4.txt | cut -c 1-330; tail -5 gpurun_out/r04_call_latency.txt | cut -c 1-200
