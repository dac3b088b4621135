"""Which reads can make the reference panic: host front end + C oracle against the Python restatement (no GPU).

The same sweep runs through both front ends on the GPU in tests/test_gpu_front.py::test_error_class_sweep.
"""
import pytest

from inquistr_amd import call, hipcall
from inquistr_amd.batch import INQ_READ_IS_2D, INQ_READ_SA_PANIC
from oracle import pyoracle as py
from tests import errclass


def test_sweep_covers_both_outcomes():
    n_panic = {False: 0, True: 0}
    n = 0
    for _name, probe in errclass.cases():
        n += 1
        for unphased in (False, True):
            n_panic[unphased] += errclass.expected(errclass.good_reads() + [probe], unphased) is None
    assert n == 2 * 4 * 3 * 2 * 10
    # unphased: only a kept (spanning, mapq 60) read with a clip and a panicking SA; phased: HP 's' on any probe, or
    # a kept (HP C/i, mapq 60, not inside) read with a clip and a panicking SA
    assert n_panic[True] == 4 * 6 and n_panic[False] == 2 * 3 * 2 * 10 + 2 * 2 * 6


@pytest.mark.parametrize("unphased", [False, True])
def test_host_front_end_and_oracle_agree_with_restatement(tmp_path, orc, unphased):
    bam = str(tmp_path / "e.bam")
    n_panic = n_bit = 0
    for name, probe in errclass.cases():
        recs = errclass.write_bam(bam, errclass.good_reads() + [probe])
        want = errclass.expected(recs, unphased)
        fe = call.FrontEnd(bam, region="%s:%d-%d" % errclass.LOCUS, unphased=unphased)
        try:
            (batch, _idx), = list(fe.batches())
        except call.CallError as e:  # get_phase on a fetched read, before any filter (src/call.rs:349)
            assert e.status == 101 and want is None and probe.hp == ("s", 1) and not unphased, name
            n_panic += 1
            continue
        finally:
            fe.close()
        code, res = orc.call_batch(batch)
        bits = int(batch.reads["bits"][-1])  # the probe is the last record in file order (stable sort, pos >= 4800)
        n_bit += bool(bits & INQ_READ_SA_PANIC)
        assert not (bits & INQ_READ_SA_PANIC and bits & INQ_READ_IS_2D), name
        if want is None:
            assert code == hipcall.INQ_ERR_AUX, (name, code)
            n_panic += 1
        else:
            assert code == 0, (name, code)
            assert py.format_row(*errclass.LOCUS, res.phase1[0], res.phase2[0]) == want, name
    assert n_panic > 20 and n_bit > n_panic / 4
