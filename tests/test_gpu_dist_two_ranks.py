"""Two (and three) REAL ranks driving the HIP kernels through floydwarshall_amd.dist on ONE GPU.

RCCL refuses two ranks on one device, so the process group here is gloo (which moves CUDA tensors
through host memory): everything else -- row partition, snapshot panels on a side stream,
look-ahead, sub-slab launches, the global max-form domain vote -- is exactly the production path
that `bench.py --gpus N` runs over RCCL.  Result must equal the oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, block, engine_name, kind, with_next, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from floydwarshall_amd import dist as fwdist
    from floydwarshall_amd import synth
    rate, nxt, _ = synth.make(kind, n, np.float32, seed=777)
    b = fwdist.row_bounds(n, world)
    dev = torch.device("cuda:0")
    slab = torch.from_numpy(rate[b[rank]:b[rank + 1]].copy()).to(dev)
    nslab = torch.from_numpy(nxt[b[rank]:b[rank + 1]].copy()).to(dev) if with_next else None
    backend = fwdist.HipBackend(engine_name)
    fwdist.solve_partitioned(slab, n, rank, world, nxt=nslab, block=block, backend=backend)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "rate_%d.npy" % rank), slab.cpu().numpy())
    if with_next:
        np.save(os.path.join(outdir, "next_%d.npy" % rank), nslab.cpu().numpy())
    with open(os.path.join(outdir, "nonneg_%d.txt" % rank), "w") as f:
        f.write(str(int(backend.nonneg)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n,block,engine_name,kind,with_next", [
    (2, 512, 64, "fused", "d1", False),     # max-form domain holds on every rank
    (2, 512, 64, "fused", "t3", False),     # NaN / negative on some rank: everyone must fall back
    (2, 516, 48, "fused", "t1", True),      # ties + next-hop matrix, ragged panels
    (3, 384, 64, "perk", "d2", True),       # per-k backend, three ranks
])
def test_two_ranks_one_gpu(tmp_path, world, n, block, engine_name, kind, with_next):
    import torch.multiprocessing as mp
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal
    mp.spawn(_worker, args=(world, _free_port(), n, block, engine_name, kind, with_next,
                            str(tmp_path)), nprocs=world, join=True)
    rate, nxt, _ = synth.make(kind, n, np.float32, seed=777)
    oracle.relax(rate, nxt if with_next else None)
    got = np.concatenate([np.load(tmp_path / ("rate_%d.npy" % r)) for r in range(world)])
    assert_bits_equal(got, rate, "partitioned rate")
    if with_next:
        gn = np.concatenate([np.load(tmp_path / ("next_%d.npy" % r)) for r in range(world)])
        assert_bits_equal(gn, nxt, "partitioned next")
    votes = {open(tmp_path / ("nonneg_%d.txt" % r)).read() for r in range(world)}
    assert len(votes) == 1                                       # the domain vote is global
    if engine_name == "fused" and not with_next:
        assert votes == ({"1"} if kind == "d1" else {"0"})


def _rccl_worker(rank, port, n, engine_name, with_next, outdir):
    import datetime
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=120))
    from floydwarshall_amd import dist as fwdist
    from floydwarshall_amd import synth
    rate, nxt, _ = synth.make("d2", n, np.float32, seed=778)
    slab = torch.from_numpy(rate).to(dev)
    nslab = torch.from_numpy(nxt).to(dev) if with_next else None
    fwdist.solve_partitioned(slab, n, 0, 1, nxt=nslab, backend=fwdist.HipBackend(engine_name),
                             force_collectives=True)
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # what bench.py does with the step time
    assert float(t.item()) == 1.5
    np.save(os.path.join(outdir, "rate.npy"), slab.cpu().numpy())
    if with_next:
        np.save(os.path.join(outdir, "next.npy"), nslab.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("engine_name,with_next", [("fused", False), ("perk", True)])
def test_rccl_calls_on_a_single_rank(tmp_path, engine_name, with_next):
    """RCCL itself (backend "nccl"), one rank: the MIN all-reduce of the domain vote and the async
    panel broadcasts on the side stream are issued exactly as with N ranks (force_collectives),
    so stream ordering against RCCL's internal stream is exercised on the real library."""
    import torch.multiprocessing as mp
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal
    n = 640
    mp.spawn(_rccl_worker, args=(_free_port(), n, engine_name, with_next, str(tmp_path)),
             nprocs=1, join=True)
    rate, nxt, _ = synth.make("d2", n, np.float32, seed=778)
    oracle.relax(rate, nxt if with_next else None)
    assert_bits_equal(np.load(tmp_path / "rate.npy"), rate, "rate over RCCL")
    if with_next:
        assert_bits_equal(np.load(tmp_path / "next.npy"), nxt, "next over RCCL")
