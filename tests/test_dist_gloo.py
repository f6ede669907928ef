"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups drive
floydwarshall_amd.dist.solve_partitioned -- the row-block partition, the pivot-panel schedule, the
look-ahead and the broadcast -- with an oracle-backed CPU backend standing in for the HIP kernels
(the product backend is HIP only; this checks the schedule, not the kernels).  The gathered result
must equal the oracle's single-process solve bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleBackend:
    """relax / panel restated with numpy for slabs with an explicit snapshot panel.
    Step k on a slab (Algorithms.hs:42-61): c = r[i][k] * w_k[j]; update iff r[i][j] < c;
    skip i == k (:50) and j in {i, k} (:54); next' = next[i][k], hops' = hops[i][k] + hops[k][j]."""

    @staticmethod
    def _apply(rate, nxt, hops, n, row0, k, wk, whk):
        rows = rate.shape[0]
        with np.errstate(all="ignore"):
            c = rate[:, k:k + 1] * wk[None, :]
        mask = rate < c
        mask[:, k] = False
        gi = np.arange(row0, row0 + rows)
        mask[gi == k, :] = False
        inside = (gi >= 0) & (gi < n)
        mask[np.arange(rows)[inside], gi[inside]] = False
        if nxt is not None:
            nk = nxt[:, k].copy()
            nxt[mask] = np.broadcast_to(nk[:, None], mask.shape)[mask]
        if hops is not None:
            hsum = hops[:, k:k + 1] + whk[None, :]
            hops[mask] = hsum[mask]
        rate[mask] = c[mask]

    @staticmethod
    def _np(t):
        return None if t is None else t.numpy()

    def relax(self, slab, n, row0, k0, k1, w, wh=None):
        r, nx, hp = slab.rate.numpy(), self._np(slab.nxt), self._np(slab.hops)
        wn, whn = w.numpy(), self._np(wh)
        for k in range(k0, k1):
            self._apply(r, nx, hp, n, row0, k, wn[k - k0], None if whn is None else whn[k - k0])

    def relax_skipping(self, slab, n, row0, k0, k1, w, wh, skip):
        """One call for the whole slab minus the rows skip = (lo, hi) -- what the HIP backend does
        in a single launch per pivot; here simply the two parts.  Multiples of 4 only (as HIP)."""
        lo, hi = skip
        if lo % 4 or hi % 4:
            return False
        self.skip_calls = getattr(self, "skip_calls", 0) + 1
        self.relax(slab.rows(0, lo), n, row0, k0, k1, w, wh)
        self.relax(slab.rows(hi, slab.nrows), n, row0 + hi, k0, k1, w, wh)
        return True

    def panel(self, block, n, k0, w, wh=None):
        r = block.rate.numpy().copy()         # snapshot only: the matrix itself is not modified
        hp = None if block.hops is None else block.hops.numpy().copy()
        wn, whn = w.numpy(), self._np(wh)
        for t in range(r.shape[0]):
            wn[t] = r[t]                      # time-k snapshot of pivot row k0+t
            if hp is not None:
                whn[t] = hp[t]
            self._apply(r, None, hp, n, k0, k0 + t, wn[t], None if hp is None else whn[t])


def _worker(rank, world, port, n, block, lookahead, kind, with_next, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from floydwarshall_amd import dist as fwdist
    from floydwarshall_amd import synth
    rate, nxt, hops = synth.make(kind, n, np.float32, seed=4242)
    b = fwdist.row_bounds(n, world)
    slab = torch.from_numpy(rate[b[rank]:b[rank + 1]].copy())
    nslab = torch.from_numpy(nxt[b[rank]:b[rank + 1]].copy()) if with_next else None
    hslab = torch.from_numpy(hops[b[rank]:b[rank + 1]].copy()) if with_next else None
    # (with the per-step spans the benchmark's N > 1 line reports -- wall clock on the CPU; they must not
    #  change the schedule)
    timer = fwdist.StepTimer(on_gpu=False)
    fwdist.solve_partitioned(slab, n, rank, world, nxt=nslab, hops=hslab, block=block,
                             backend=OracleBackend(), lookahead=lookahead, timer=timer)
    sm = timer.summary()
    blocks = fwdist.pivot_blocks(n, world, block)
    mine = sum(1 for i, bk in enumerate(blocks) if bk[2] == rank)
    assert sm["bulk"][1] == len(blocks) and sm["exchange"][1] == len(blocks) and sm["panel"][1] == mine
    assert sm["lookahead"][1] == (sum(1 for i, bk in enumerate(blocks) if i > 0 and bk[2] == rank) if lookahead else 0)
    assert all(v[0] >= 0.0 for v in sm.values())
    np.save(os.path.join(outdir, "rate_%d.npy" % rank), slab.numpy())
    if with_next:
        np.save(os.path.join(outdir, "next_%d.npy" % rank), nslab.numpy())
        np.save(os.path.join(outdir, "hops_%d.npy" % rank), hslab.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n,block,lookahead,kind", [
    (2, 96, 16, True, "d1"),
    (2, 96, 16, False, "d1"),
    (2, 70, 8, True, "t1"),      # ties: earliest pivot must win on every rank
    (3, 50, 7, True, "t3"),      # ragged partition, ragged panels, inf/NaN inputs
    (8, 131, 5, True, "t2"),     # the 8-rank shape of `bench.py --gpus 8`, ragged everything
    (2, 128, 16, True, "d2"),    # aligned panels: the owner takes the single skip-launch path
])
def test_partitioned_solve_equals_single_process_oracle(tmp_path, world, n, block, lookahead, kind):
    import oracle
    from floydwarshall_amd import synth
    from helpers import assert_bits_equal, spawn_ranks
    spawn_ranks(_worker, (world, _free_port(), n, block, lookahead, kind, True, str(tmp_path)), world)
    rate, nxt, hops = synth.make(kind, n, np.float32, seed=4242)
    oracle.relax(rate, nxt, hops)
    got_r = np.concatenate([np.load(tmp_path / ("rate_%d.npy" % r)) for r in range(world)])
    got_n = np.concatenate([np.load(tmp_path / ("next_%d.npy" % r)) for r in range(world)])
    got_h = np.concatenate([np.load(tmp_path / ("hops_%d.npy" % r)) for r in range(world)])
    assert_bits_equal(got_r, rate, "partitioned rate")
    assert_bits_equal(got_n, nxt, "partitioned next")
    assert_bits_equal(got_h, hops, "partitioned hops (they travel with the panels)")


def test_partition_helpers():
    from floydwarshall_amd import dist as fwdist
    assert fwdist.row_bounds(10, 3) == [0, 3, 6, 10]
    blocks = fwdist.pivot_blocks(10, 3, 2)
    assert [b[0] for b in blocks] == [0, 2, 3, 5, 6, 8]
    assert sum(b[1] for b in blocks) == 10
    for k0, b, owner in blocks:       # a panel never straddles two owners
        lo, hi = fwdist.row_bounds(10, 3)[owner], fwdist.row_bounds(10, 3)[owner + 1]
        assert lo <= k0 and k0 + b <= hi
    assert fwdist.pivot_blocks(0, 2, 4) == []
