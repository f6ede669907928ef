#!/bin/bash
# round 4, run 11: the torchrun form of the N > 1 bench rehearsed with 2 and 4 ranks on one GPU over gloo: the
# timed per-k path (dist.solve_partitioned), event timings, and the fused leg through dist.PartMatrix (pair schedule)
cd "$GRAFT_REPO_ROOT"
for w in 2 4; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2951$w bench.py --gpus $w --backend gloo --size 8192 --steps 1 --warmup 1 --cpu-seconds 2 > gpurun_out/r04_bench_dist_rehearsal_${w}ranks_gloo.txt 2> gpurun_out/r04_bench_dist_rehearsal_${w}ranks.err; echo "dist rehearsal $w ranks rc=$?"
grep '^{' gpurun_out/r04_bench_dist_rehearsal_${w}ranks_gloo.txt > gpurun_out/r04_bench_dist_rehearsal_${w}ranks_gloo.json
python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_dist_rehearsal_${w}ranks_gloo.json"))
print(d["n_gpus"], d["value"], d["ms_per_step"], {k:v for k,v in d["exchange"].items() if k!="timing_note"})
print(d.get("fused_engine"))
print(d.get("extras_aborted"), d.get("INVALID_rehearsal_backend"))
PY
grep -v Warning gpurun_out/r04_bench_dist_rehearsal_${w}ranks.err | tail -5
done
