#!/bin/bash
# rehearsal of the N > 1 benchmark path on ONE GPU: 2 and 3 ranks over gloo (never a performance number)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for n in 2 3; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 1 --warmup 1 --size 4096 --backend gloo > $O/r02_run10_rehearse$n.json 2> $O/r02_run10_rehearse$n.err || { tail -20 $O/r02_run10_rehearse$n.err; exit 1; }
  tail -1 $O/r02_run10_rehearse$n.json | cut -c1-900
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --size 4096 --backend gloo --engine fused --with-next > $O/r02_run10_rehearse2f.json 2> $O/r02_run10_rehearse2f.err || { tail -20 $O/r02_run10_rehearse2f.err; exit 1; }
tail -1 $O/r02_run10_rehearse2f.json | cut -c1-600
if grep -l "Memory access fault" $O/r02_run10_* 2>/dev/null; then echo "GPU FAULT"; exit 9; fi
