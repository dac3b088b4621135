"""CPU-side checks: the C-ABI library loads and exports every declared symbol, fails loudly
without a GPU, the batch layout matches the header, the synthetic generator is deterministic,
and the committed fixture agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from inquistr_amd import batch as B
from inquistr_amd import hipcall, synth
from tests import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "inquistr_hip.h")).read()
    declared = set(re.findall(r"\b(inq_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(hipcall.ABI_SYMBOLS)
    L = hipcall.load()
    for s in declared:
        assert hasattr(L, s), s
    assert L.inq_abi_version() == 5
    assert b"no CPU fallback" in L.inq_strerror(B.INQ_ERR_NO_DEVICE)


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    with pytest.raises(hipcall.InqError) as e:
        hipcall.Context(0)
    assert e.value.code == B.INQ_ERR_NO_DEVICE


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_list_form_fails_loudly_without_gpu_and_leaves_nothing_behind():
    with pytest.raises(hipcall.InqError) as e:
        hipcall.Context.create_multi([0, 1])
    assert e.value.code == B.INQ_ERR_NO_DEVICE
    L = hipcall.load()
    import ctypes as C

    hs = (C.c_void_p * 2)(1, 1)
    assert L.inq_ctx_create_multi((C.c_int * 2)(0, 0), 2, hs) == B.INQ_ERR_NO_DEVICE and not hs[0] and not hs[1]
    assert L.inq_ctx_create_multi(None, 2, hs) == B.INQ_ERR_ARG and L.inq_ctx_create_multi((C.c_int * 2)(0, 0), 0, hs) == B.INQ_ERR_ARG


def test_struct_layout_matches_header():
    assert B.READ_DTYPE.itemsize == 16
    assert C.sizeof(B.InqBatchC) == 4 * 8 + 6 * 8 + 4 * 4
    assert C.sizeof(B.InqResultC) == 5 * 8
    hdr = open(os.path.join(ROOT, "include", "inquistr_hip.h")).read()
    assert C.sizeof(hipcall.BgzfBlockC) == 24 == hipcall.BGZF_BLOCK_DTYPE.itemsize
    assert C.sizeof(hipcall.SpanC) == 11 * 8 + 4 * 4 and C.sizeof(hipcall.SpanStatsC) == 5 * 8 + 2 * 4 + 8 + 5 * 8
    for name, val in (("INQ_ERR_ARG", -1), ("INQ_ERR_INDEX", -7), ("INQ_ERR_NO_DEVICE", -10), ("INQ_ERR_INFLATE", -11),
                      ("INQ_ERR_BAM", -12), ("INQ_ERR_AUX", -13)):
        assert re.search(rf"{name} = {val}\b", hdr) and getattr(B, name) == val


def test_builder_pads_reads_to_four_words():
    bb = B.BatchBuilder()
    a = bb.add_read(10, B.encode_cigar([("M", 5)] * 5), phase=1)
    b = bb.add_read(20, B.encode_cigar([("M", 7)] * 4), phase=None)
    bb.add_locus(100, 200, [a, b])
    batch = bb.build()
    assert batch.cigar.shape[0] == 12 and list(batch.reads["cigar_off4"]) == [0, 2]
    assert list(batch.cigar[5:8]) == [0, 0, 0]
    assert batch.reads["bits"][0] == B.INQ_READ_HAS_HP and batch.reads["bits"][1] == 0
    assert batch.algorithmic_bytes() == 4 * 9 + 20 * 2 + 32


def test_slice_loci_keeps_results(orc):
    batch, _ = gen.random_case(5, n_loci=30, unphased=True)
    _, full = orc.call_batch(batch)
    sub = batch.slice_loci(7, 19)
    _, part = orc.call_batch(sub)
    assert gen.same_f64(part.phase1, full.phase1[7:19]) and gen.same_f64(part.phase2, full.phase2[7:19])


def test_synth_deterministic_and_sliceable(orc):
    wl = synth.WORKLOADS["unphased100k"]
    a = synth.generate_numpy(wl, 1000, 1040)
    b = synth.generate_numpy(wl, 1000, 1040)
    assert np.array_equal(a.cigar, b.cigar) and np.array_equal(a.reads, b.reads)
    c = synth.generate_numpy(wl, 1010, 1020)
    _, ra = orc.call_batch(a)
    _, rc = orc.call_batch(c)
    assert gen.same_f64(ra.phase1[10:20], rc.phase1) and gen.same_f64(ra.phase2[10:20], rc.phase2)
    ops = a.cigar_ops_per_pair()
    assert 180 <= ops.min() and ops.max() <= 220 and a.n_pairs == 40 * 30
    # reads really cover their window and every locus is callable
    assert not np.isnan(ra.phase1).any() and not np.isnan(ra.phase2).any()


def test_synth_torch_matches_numpy():
    wl = synth.WORKLOADS["phased10k"]
    d = synth.DeviceBatch(wl, "cpu", 50, 120)
    b = synth.generate_numpy(wl, 50, 120)
    assert np.array_equal(d.cigar.numpy().view(np.uint32), b.cigar)
    assert np.array_equal(d.reads.numpy().reshape(-1).view(B.READ_DTYPE), b.reads)
    assert d.algorithmic_bytes() == b.algorithmic_bytes()


def test_committed_fixture_matches_oracle(orc):
    z = np.load(os.path.join(ROOT, "tests", "golden", "random_batches.npz"))
    for i in range(int(z["n_cases"])):
        batch = B.Batch(
            cigar=z[f"c{i}_cigar"], reads=z[f"c{i}_reads"].view(B.READ_DTYPE).reshape(-1),
            pair_read=z[f"c{i}_pair_read"], locus_pair_off=z[f"c{i}_off"], locus_start=z[f"c{i}_start"],
            locus_end=z[f"c{i}_end"], minlen=int(z[f"c{i}_params"][0]), support=int(z[f"c{i}_params"][1]),
            unphased=bool(z[f"c{i}_params"][2]),
        )
        code, res = orc.call_batch(batch, debug=True)
        assert code == 0
        assert gen.same_f64(res.phase1, z[f"c{i}_p1"]) and gen.same_f64(res.phase2, z[f"c{i}_p2"])
        assert np.array_equal(res.pair_call, z[f"c{i}_pair_call"])
        assert np.array_equal(res.pair_bits, z[f"c{i}_pair_bits"])
        assert res.n_tie_loci == int(z[f"c{i}_params"][3])


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
@pytest.mark.parametrize("frontend", ["host", "device", None])
def test_whole_command_fails_loudly_without_gpu(tmp_path, frontend):
    """Neither front end has a CPU path for the calling: without a gfx950 device the command ends with status 1."""
    from inquistr_amd import call
    from tests.test_host_frontend import _make_case

    bam, bed, loci, recs = _make_case(tmp_path, 3, n_loci=10)
    with pytest.raises(call.CallError) as e, open(tmp_path / "o.inq", "w") as f:
        call.genotype_repeats(bam, None, bed, 5, 3, 2, False, None, None, out=f, frontend=frontend)
    assert e.value.status == 1 and "no CPU fallback" in e.value.message
    assert (tmp_path / "o.inq").read_text() == ""


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_session_and_run_fail_loudly_without_gpu(tmp_path):
    """The round-3 entries (many BAMs on one context: inq_session_*; a prepared run: inq_run_rows) have no CPU path either: every
    file of a cohort ends with status 1, nothing is written; the parts that need no GPU (split, output stage) still work."""
    from inquistr_amd import call
    from tests.test_host_frontend import _make_case

    bam, bed, loci, recs = _make_case(tmp_path, 4, n_loci=12)
    outs = [open(tmp_path / f"s{k}.inq", "w") for k in range(3)]
    with call.Session(0) as S:
        st = S.call_many([bam, str(tmp_path / "missing.bam"), bam], outs, region_file=bed, threads=2)
        assert st == [1, 1, 1] and "no CPU fallback" in S.last_message
        with pytest.raises(call.CallError) as e, open(tmp_path / "one.inq", "w") as f:
            S.call(bam, region_file=bed, out=f)
        assert e.value.status == 1
    for o in outs:
        o.close()
    assert all((tmp_path / f"s{k}.inq").read_text() == "" for k in range(3))
    run = call.Run(bam, None, bed, threads=2, sample_name="S")
    order, cuts = run.partition(3)
    assert sorted(order.tolist()) == list(range(run.n_targets)) and cuts[0] == 0 and cuts[-1] == run.n_targets
    with pytest.raises(call.CallError) as e:
        run.rows(order[:4])
    assert e.value.status == 1
    with open(tmp_path / "rows.inq", "w") as f:  # the output stage alone: NaN rows in -t 2 order
        run.write_inq(np.full(run.n_targets, np.nan), np.full(run.n_targets, np.nan), f)
    text = (tmp_path / "rows.inq").read_text().splitlines()
    assert text[0] == "chromosome\tbegin\tend\tS_H1\tS_H2" and len(text) == run.n_targets + 1 and text[1].endswith("\tNaN\tNaN")
    run.close()
    # a run opened ON a session (inq_session_run_open: what a resident rank of call_dist uses): the same split, the same loud failure
    with call.Session(0) as S:
        srun = call.Run(bam, None, bed, threads=2, sample_name="S", session=S)
        o2, c2 = srun.partition(3)
        assert np.array_equal(o2, order) and np.array_equal(c2, cuts)
        for _ in range(2):  # (the session survives the failing call)
            with pytest.raises(call.CallError) as e:
                srun.rows(order[:4])
            assert e.value.status == 1 and "no CPU fallback" in e.value.message
        with pytest.raises(call.CallError) as e:
            srun.rows_device(order[:4], 8)
        assert e.value.status == 1
        # (left open: the session closes the runs still open on it before it goes)
    assert not (srun._h and srun._h.value)
    # call_dist (one process per GPU) takes the process's session by itself; a rank whose device work ends with an error exit gives the
    # session up, so that the next call makes a new context instead of meeting the dead one
    from inquistr_amd import call_dist

    for _ in range(2):
        with pytest.raises(call.CallError) as e, open(tmp_path / "dist.inq", "w") as f:
            call_dist.genotype_repeats_distributed(bam, None, bed, 5, 3, 2, False, "S", out=f)
        assert e.value.status == 1 and "no CPU fallback" in e.value.message
        assert call_dist._sessions == {}
    assert (tmp_path / "dist.inq").read_text() == ""
