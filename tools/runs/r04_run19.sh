#!/bin/bash
# round 4, run 19: the torchrun form of the bench with 2 ranks on one GPU over gloo, small matrix (the ranks
# time-slice the GPU: a rehearsal of the code path, never a performance number)
cd "$GRAFT_REPO_ROOT"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --size 4096 --steps 2 --warmup 1 --cpu-seconds 2 > gpurun_out/r04_bench_dist_rehearsal_2ranks_gloo.txt 2> gpurun_out/r04_bench_dist_rehearsal_2ranks.err; echo "rc=$?"
grep '^{' gpurun_out/r04_bench_dist_rehearsal_2ranks_gloo.txt > gpurun_out/r04_bench_dist_rehearsal_2ranks_gloo.json
python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_dist_rehearsal_2ranks_gloo.json"))
print(d["n_gpus"], d["value"], d["ms_per_step"], d["driver"]["name"]); print(d["exchange"]); print(d.get("fused_engine")); print(d.get("legacy_driver")); print(d.get("extras_aborted"))
PY
grep -v "Warning\|warn\|check(lib" gpurun_out/r04_bench_dist_rehearsal_2ranks.err | tail -4
