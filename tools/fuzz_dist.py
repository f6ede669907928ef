#!/usr/bin/env python3
"""Developer fuzz of the partitioned solve: 2-3 real processes on ONE GPU over gloo, HIP kernels,
random sizes / blocks / engines / input kinds, against the oracle.  usage: fuzz_dist.py [cases]"""
import os
import socket
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, n, block, engine_name, kind, with_next, seed, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from floydwarshall_amd import dist as fwdist
    from floydwarshall_amd import synth
    rate, nxt, _ = synth.make(kind, n, np.float32, seed=seed)
    b = fwdist.row_bounds(n, world)
    dev = torch.device("cuda:0")
    slab = torch.from_numpy(rate[b[rank]:b[rank + 1]].copy()).to(dev)
    nslab = torch.from_numpy(nxt[b[rank]:b[rank + 1]].copy()).to(dev) if with_next else None
    fwdist.solve_partitioned(slab, n, rank, world, nxt=nslab, block=block,
                             backend=fwdist.HipBackend(engine_name))
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "rate_%d.npy" % rank), slab.cpu().numpy())
    if with_next:
        np.save(os.path.join(outdir, "next_%d.npy" % rank), nslab.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rnd = np.random.default_rng(12345)
    for c in range(cases):
        world = int(rnd.integers(2, 4))
        n = 4 * int(rnd.integers(40, 260))
        block = int(rnd.choice([8, 16, 24, 48, 64]))
        engine_name = str(rnd.choice(["fused", "perk"]))
        kind = str(rnd.choice(["d1", "d2", "t1", "t2", "t3"]))
        with_next = bool(rnd.integers(0, 2))
        seed = int(rnd.integers(0, 10000))
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker, args=(world, free_port(), n, block, engine_name, kind, with_next, seed, d),
                     nprocs=world, join=True)
            rate, nxt, _ = synth.make(kind, n, np.float32, seed=seed)
            oracle.relax(rate, nxt if with_next else None)
            got = np.concatenate([np.load(os.path.join(d, "rate_%d.npy" % r)) for r in range(world)])
            assert_bits_equal(got, rate, "rate")
            if with_next:
                gn = np.concatenate([np.load(os.path.join(d, "next_%d.npy" % r)) for r in range(world)])
                assert_bits_equal(gn, nxt, "next")
        print("case %d ok: world=%d n=%d block=%d %s %s next=%s" % (c, world, n, block, engine_name, kind, with_next),
              flush=True)
    print("fuzz_dist: OK")
