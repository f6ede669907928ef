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
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal, spawn_ranks
    spawn_ranks(_worker, (world, _free_port(), n, block, engine_name, kind, with_next, str(tmp_path)),
                world)
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
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal, spawn_ranks
    n = 640
    spawn_ranks(_rccl_worker, (_free_port(), n, engine_name, with_next, str(tmp_path)), 1)
    rate, nxt, _ = synth.make("d2", n, np.float32, seed=778)
    oracle.relax(rate, nxt if with_next else None)
    assert_bits_equal(np.load(tmp_path / "rate.npy"), rate, "rate over RCCL")
    if with_next:
        assert_bits_equal(np.load(tmp_path / "next.npy"), nxt, "next over RCCL")


def _driver_worker(rank, n, engine_name, lookahead, outdir):
    """solve_partitioned at world size 1 (no process group): rate + next, then rate + next + hops
    (+ the path trace on the fused engine) -- config 5's fields through the driver."""
    import warnings
    import torch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)     # libfwx on torch's HIP runtime: expected here
        from floydwarshall_amd import dist as fwdist
        from floydwarshall_amd import engine, synth
    dev = torch.device("cuda:0")
    rate, nxt, _ = synth.make("t2", n, np.float32, seed=41)
    r, nx = torch.from_numpy(rate).to(dev), torch.from_numpy(nxt).to(dev)
    fwdist.solve_partitioned(r, n, 0, 1, nxt=nx, block=48, lookahead=lookahead,
                             backend=fwdist.HipBackend(engine_name))
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "a_rate.npy"), r.cpu().numpy())
    np.save(os.path.join(outdir, "a_next.npy"), nx.cpu().numpy())
    rate, nxt, hops = synth.make("t1", n, np.float32, seed=43)
    r, nx, hp = (torch.from_numpy(a.copy()).to(dev) for a in (rate, nxt, hops))
    trace = engine.Trace(n, n, dev) if engine_name == "fused" else None
    fwdist.solve_partitioned(r, n, 0, 1, nxt=nx, hops=hp, trace=trace, block=64, lookahead=lookahead,
                             backend=fwdist.HipBackend(engine_name))
    torch.cuda.synchronize()
    for name, t in (("b_rate", r), ("b_next", nx), ("b_hops", hp)):
        np.save(os.path.join(outdir, name + ".npy"), t.cpu().numpy())
    if trace is not None:
        for name in ("last", "at_col", "at_row"):
            np.save(os.path.join(outdir, name + ".npy"), getattr(trace, name).cpu().numpy())


@pytest.mark.parametrize("engine_name", ["fused", "perk"])
@pytest.mark.parametrize("lookahead", [True, False])
def test_dist_driver_single_rank_on_gpu(tmp_path, engine_name, lookahead):
    """floydwarshall_amd.dist.solve_partitioned with the HIP backend at world size 1: the panel /
    look-ahead schedule drives the real kernels on torch-owned memory (a process of its own: the
    driver needs torch, this one stays torch-free)."""
    import oracle
    from floydwarshall_amd import engine, synth
    from helpers import assert_bits_equal, path_from_trace, spawn_ranks
    n = 448
    spawn_ranks(_driver_worker, (n, engine_name, lookahead, str(tmp_path)), 1)
    rate, nxt, _ = synth.make("t2", n, np.float32, seed=41)
    oracle.relax(rate, nxt)
    assert_bits_equal(np.load(tmp_path / "a_rate.npy"), rate, "rate")
    assert_bits_equal(np.load(tmp_path / "a_next.npy"), nxt, "next")
    rate, nxt, hops = synth.make("t1", n, np.float32, seed=43)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    oracle.relax(er, en, eh)
    assert_bits_equal(np.load(tmp_path / "b_rate.npy"), er, "rate")
    assert_bits_equal(np.load(tmp_path / "b_next.npy"), en, "next")
    assert_bits_equal(np.load(tmp_path / "b_hops.npy"), eh, "hops")
    if engine_name == "fused":
        last, at_col, at_row = (np.load(tmp_path / (f + ".npy")) for f in ("last", "at_col", "at_row"))
        rnd = np.random.default_rng(9)
        with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt)
            dm.solve()
            for _ in range(200):
                a, b = int(rnd.integers(0, n)), int(rnd.integers(0, n))
                assert path_from_trace(last, at_col, at_row, nxt, a, b) == dm.query_exact(a, b)[1]
