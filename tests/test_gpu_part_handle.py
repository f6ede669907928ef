"""One partition per PROCESS behind the C ABI (fwx_matrix_create_part, round 4): the partitioned handle's own
schedules -- single pass, the 128-pivot pair schedule, the per-k engine, recording / resumed solves -- with
the panel exchange handed to the host (floydwarshall_amd.dist.PartMatrix: torch.distributed.broadcast from
libfwx's callback, on the partition's side stream).  RCCL refuses two ranks on one device, so the ranks here
talk gloo (CUDA tensors through host memory); everything else is the production path of
`torchrun ... bench.py --gpus N`.  Every rank's slab against the oracle, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port, env):
    import warnings
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for k, v in env.items():
        os.environ[k] = v
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)     # libfwx on torch's HIP runtime: expected in a rank
        from floydwarshall_amd import dist as fwdist
        from floydwarshall_amd import engine, synth
        engine.device_count()
    return dist, fwdist, engine, synth


def _solve_worker(rank, world, port, n, dtype_name, kind, fields, eng_name, env, outdir):
    dist, fwdist, engine, synth = _init(rank, world, port, env)
    dtype = np.dtype(dtype_name).type
    rate, nxt, hops = synth.make(kind, n, dtype, seed=4242)
    with_next, with_hops, traced = fields >= 1, fields >= 2, fields >= 3
    h = fwdist.PartMatrix(n, dtype, rank, world, with_next=with_next, with_hops=with_hops, device=0)
    lo, hi = h.row0, h.row0 + h.rows            # where the LIBRARY cut the matrix (64-aligned for n >= 128 * world)
    assert h.bounds()[rank] == lo and h.bounds()[-1] == n
    if traced:
        h.enable_path_log()
    h.set_timing(True)
    cut = lambda a, on: np.ascontiguousarray(a[lo:hi]) if on else None  # noqa: E731
    h.upload(cut(rate, True), cut(nxt, with_next), cut(hops, with_hops))
    eng = {"fused": engine.FWX_ENGINE_FUSED, "perk": engine.FWX_ENGINE_PERK, "auto": engine.FWX_ENGINE_AUTO}[eng_name]
    u = h.solve(engine=eng, count_updates=(eng_name == "perk"))
    t = h.timing()
    gr, gn, gh = h.download()
    np.save(os.path.join(outdir, "rate_%d.npy" % rank), gr)
    if with_next:
        np.save(os.path.join(outdir, "next_%d.npy" % rank), gn)
    if with_hops:
        np.save(os.path.join(outdir, "hops_%d.npy" % rank), gh)
    with open(os.path.join(outdir, "meta_%d.txt" % rank), "w") as f:
        f.write("%d %d %d" % (u or 0, t["pivots_per_step"], t["steps"]))
    h.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,dtype_name,kind,fields,eng_name,pairs", [
    (2, 512, "float32", "d1", 0, "fused", False),      # rates only, single pass (below the pair threshold)
    (2, 640, "float32", "t1", 1, "fused", True),       # ties + next-hops, pair schedule forced
    (4, 1024, "float64", "d2", 2, "fused", True),      # f64 + next + hops over four ranks, pairs
    (3, 576, "float32", "t2", 3, "auto", True),        # sparse + the path trace, pairs straddling owners
    (3, 500, "float64", "d1", 1, "fused", True),       # n no multiple of anything: aligned partitions (0, 128, 320), pairs
    (3, 300, "float64", "d1", 1, "fused", False),      # n < 128 * world: balanced, unaligned partitions (0, 100, 200)
    (2, 512, "float32", "d2", 2, "perk", False),       # the per-k engine, U per rank
])
def test_one_partition_per_process_equals_the_oracle(tmp_path, world, n, dtype_name, kind, fields, eng_name, pairs):
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal, spawn_ranks
    env = {"FWX_DOUBLE_PASS_MIN_N": "0", "FWX_DOUBLE_PASS_NEXT_MIN_N": "0"} if pairs else \
        {"FWX_DOUBLE_PASS_MIN_N": "100000000", "FWX_DOUBLE_PASS_NEXT_MIN_N": "100000000"}
    spawn_ranks(_solve_worker, (world, _free_port(), n, dtype_name, kind, fields, eng_name, env, str(tmp_path)), world)
    dtype = np.dtype(dtype_name).type
    rate, nxt, hops = synth.make(kind, n, dtype, seed=4242)
    eu = oracle.relax(rate, nxt if fields >= 1 else None, hops if fields >= 2 else None)
    cat = lambda name: np.concatenate([np.load(tmp_path / ("%s_%d.npy" % (name, r))) for r in range(world)])  # noqa: E731
    assert_bits_equal(cat("rate"), rate, "rate")
    if fields >= 1:
        assert_bits_equal(cat("next"), nxt, "next")
    if fields >= 2:
        assert_bits_equal(cat("hops"), hops, "hops")
    meta = [[int(x) for x in open(tmp_path / ("meta_%d.txt" % r)).read().split()] for r in range(world)]
    if eng_name == "perk":
        assert sum(m[0] for m in meta) == eu                       # every rank reports its share of U
    assert {m[1] for m in meta} == {128 if pairs else 64}          # the schedule under test, on every rank
    assert all(m[2] >= 1 for m in meta)


def _resume_worker(rank, world, port, n, outdir):
    dist, fwdist, engine, synth = _init(rank, world, port, {})
    rate, nxt, _ = synth.make("d2", n, np.float32, seed=99)
    h = fwdist.PartMatrix(n, np.float32, rank, world, with_next=True, device=0)
    lo, hi = h.row0, h.row0 + h.rows
    h.keep_input()
    placed = h.enable_resume(3)
    h.upload(np.ascontiguousarray(rate[lo:hi]), np.ascontiguousarray(nxt[lo:hi]))
    h.solve()
    started = []
    for step, (u, v) in enumerate([(400, 300), (130, 500), (40, 90)]):
        idx = np.array([u * n + v, v * n + u], dtype=np.int64)        # the caller's n x n indices, on every rank
        vals = (rate.reshape(-1)[idx] * np.float32(0.9 + 0.02 * step)).astype(np.float32)
        rate.reshape(-1)[idx] = vals
        started.append(h.resolve(idx, vals, np.array([v, u], dtype=np.int32)))
        gr, gn, _ = h.download()
        np.save(os.path.join(outdir, "rate_%d_%d.npy" % (step, rank)), gr)
        np.save(os.path.join(outdir, "next_%d_%d.npy" % (step, rank)), gn)
    with open(os.path.join(outdir, "meta_%d.txt" % rank), "w") as f:
        f.write("%d %s" % (placed, " ".join(map(str, started))))
    h.close()
    dist.barrier()
    dist.destroy_process_group()


def test_resumed_solves_across_processes(tmp_path):
    """keep_input + enable_resume + resolve on two ranks: every rank gets the same entry indices and applies
    its rows'; the solve restarts at the same checkpoint everywhere; results as a from-scratch oracle solve."""
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal, spawn_ranks
    world, n = 2, 512
    spawn_ranks(_resume_worker, (world, _free_port(), n, str(tmp_path)), world)
    rate, nxt, _ = synth.make("d2", n, np.float32, seed=99)
    for step, (u, v) in enumerate([(400, 300), (130, 500), (40, 90)]):
        idx = np.array([u * n + v, v * n + u], dtype=np.int64)
        rate.reshape(-1)[idx] = (rate.reshape(-1)[idx] * np.float32(0.9 + 0.02 * step)).astype(np.float32)
        er, en = rate.copy(), nxt.copy()
        oracle.relax(er, en)
        gr = np.concatenate([np.load(tmp_path / ("rate_%d_%d.npy" % (step, r))) for r in range(world)])
        gn = np.concatenate([np.load(tmp_path / ("next_%d_%d.npy" % (step, r))) for r in range(world)])
        assert_bits_equal(gr, er, "rate after change %d" % step)
        assert_bits_equal(gn, en, "next after change %d" % step)
    metas = {open(tmp_path / ("meta_%d.txt" % r)).read() for r in range(world)}
    assert metas == {"3 256 128 0"}                                  # checkpoints 128 / 256 / 384 on both ranks


def _rccl_worker(rank, port, n, outdir):
    import datetime
    import warnings
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FWX_DOUBLE_PASS_MIN_N"] = "0"
    os.environ["FWX_DOUBLE_PASS_NEXT_MIN_N"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        from floydwarshall_amd import dist as fwdist
        from floydwarshall_amd import engine, synth
    rate, nxt, hops = synth.make("d2", n, np.float32, seed=779)
    h = fwdist.PartMatrix(n, np.float32, 0, 1, with_next=True, with_hops=True, device=0)
    h.set_timing(True)
    # from DEVICE arrays, as the benchmark uploads its pristine slab
    h.upload_dev(torch.from_numpy(rate).to(dev), torch.from_numpy(nxt).to(dev), torch.from_numpy(hops).to(dev))
    h.solve(engine=engine.FWX_ENGINE_FUSED)
    t = h.timing()
    gr, gn, gh = h.download()
    np.save(os.path.join(outdir, "rate.npy"), gr)
    np.save(os.path.join(outdir, "next.npy"), gn)
    np.save(os.path.join(outdir, "hops.npy"), gh)
    with open(os.path.join(outdir, "meta.txt"), "w") as f:
        f.write("%d %d %.3f" % (t["pivots_per_step"], t["steps"], t["exchange_us"]))
    h.close()
    dist.barrier()
    dist.destroy_process_group()


def test_the_exchange_callback_over_real_rccl_with_one_rank(tmp_path):
    """Backend "nccl" = RCCL itself, one rank: every panel goes through libfwx's callback into
    torch.distributed.broadcast on the partition's side stream (torch.cuda.ExternalStream over libfwx's
    stream, torch.as_tensor over its panel buffer) exactly as with N ranks -- stream ordering against RCCL's
    internal stream on the real library, under the pair schedule."""
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal, spawn_ranks
    n = 640
    spawn_ranks(_rccl_worker, (_free_port(), n, str(tmp_path)), 1)
    rate, nxt, hops = synth.make("d2", n, np.float32, seed=779)
    oracle.relax(rate, nxt, hops)
    assert_bits_equal(np.load(tmp_path / "rate.npy"), rate, "rate over RCCL")
    assert_bits_equal(np.load(tmp_path / "next.npy"), nxt, "next over RCCL")
    assert_bits_equal(np.load(tmp_path / "hops.npy"), hops, "hops over RCCL")
    pps, steps, xus = open(tmp_path / "meta.txt").read().split()
    assert int(pps) == 128 and int(steps) >= 1 and float(xus) > 0
