"""Multi-process `call` (inquistr_amd/call_dist.py) with world_size 2 on gloo.  The per-rank compute is
the oracle here (tests may use it); what is covered is target slicing, the per-rank BAM sweeps, the
gather and the output order — against the text a single-process run must produce."""
import io
import os
import socket

import pytest
import torch.multiprocessing as mp

from tests.test_gpu_end_to_end import _expected_text
from tests.test_host_frontend import _make_case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bam, bed, unphased, threads, out_path):
    import torch.distributed as dist

    from inquistr_amd import call_dist
    from oracle import orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def compute(batch):
        code, res = orc.call_batch(batch)
        assert code == 0
        return res.phase1, res.phase2

    with open(out_path if rank == 0 else os.devnull, "w") as f:
        call_dist.genotype_repeats_distributed(bam, None, bed, 5, 3, threads, unphased, "S", out=f, rank=rank,
                                               world=world, compute=compute)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,unphased,threads", [(2, False, 1), (2, True, 4), (3, False, 2)])
def test_distributed_call_equals_single(tmp_path, orc, world, unphased, threads):
    bam, bed, loci, recs = _make_case(tmp_path, 41, n_loci=45, ultra_long=True)
    out = str(tmp_path / "dist.inq")
    mp.spawn(_worker, args=(world, _free_port(), bam, bed, unphased, threads, out), nprocs=world, join=True)
    assert open(out).read() == _expected_text(loci, recs, unphased, 5, 3, "S", threads)
