"""The N>1 path on CPU: world_size 2 (and 3, with an empty shard) over gloo.  The compute stand-in
is the oracle — allowed in tests — so what is exercised is the product's sharding, slicing and
gather code."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import gen


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, unphased, n_loci, out_path):
    import torch.distributed as dist

    from inquistr_amd import shard
    from oracle import orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch, _ = gen.random_case(seed, n_loci=n_loci, unphased=unphased, long_every=6)

    def compute(sub):
        code, res = orc.call_batch(sub)
        assert code == 0
        return res.phase1, res.phase2

    got = shard.run_sharded(batch, compute, rank, world)
    if rank == 0:
        np.savez(out_path, p1=got[0], p2=got[1])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_loci,unphased", [(2, 41, False), (2, 40, True), (3, 2, False)])
def test_sharded_equals_single(orc, tmp_path, world, n_loci, unphased):
    seed = 77
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), seed, unphased, n_loci, out), nprocs=world, join=True)
    batch, _ = gen.random_case(seed, n_loci=n_loci, unphased=unphased, long_every=6)
    _, want = orc.call_batch(batch)
    z = np.load(out)
    assert gen.same_f64(z["p1"], want.phase1) and gen.same_f64(z["p2"], want.phase2)


def test_balanced_ranges_cover_and_balance():
    from inquistr_amd import shard

    rng = np.random.default_rng(0)
    cost = rng.integers(1, 100, size=1000)
    cost[500:520] = 50_000  # a heavy region
    for world in (1, 2, 3, 8):
        r = shard.balanced_ranges(cost, world)
        assert r[0][0] == 0 and r[-1][1] == 1000 and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        loads = [cost[lo:hi].sum() for lo, hi in r]
        assert max(loads) <= cost.sum() / world + cost.max()
    assert shard.balanced_ranges(np.ones(2), 4)[-1][1] == 2
    assert shard.balanced_ranges(np.zeros(0), 2) == [(0, 0), (0, 0)]
