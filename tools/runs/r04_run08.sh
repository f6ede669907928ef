#!/bin/bash
# round 4, run 8: the whole GPU suite (timed), smoke(), the default bench line, and the torchrun form of the
# N > 1 bench rehearsed with 2 ranks on one GPU over gloo (watchdog, event timings, fused leg)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_final
( time python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r04_final/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -22 gpurun_out/r04_final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r04_final/bench_default.json 2> gpurun_out/r04_final/bench_default.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r04_final/bench_default.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --size 4096 --steps 1 --warmup 1 --cpu-seconds 2 > gpurun_out/r04_final/bench_dist_rehearsal_2ranks_gloo.json 2> gpurun_out/r04_final/bench_dist_rehearsal.err; echo "dist rehearsal rc=$?"; tail -c 1200 gpurun_out/r04_final/bench_dist_rehearsal_2ranks_gloo.json; tail -5 gpurun_out/r04_final/bench_dist_rehearsal.err
