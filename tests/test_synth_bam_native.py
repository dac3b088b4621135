"""The native synthetic-BAM writer (tools/synth_bam_writer.cc, bench plumbing) writes the same files as the Python one."""
import filecmp

import pytest

from tools import make_synth_bam


@pytest.mark.parametrize("workload,n_loci", [("unphased100k", 700), ("phased10k", 130), ("expansion50k", 40)])
def test_native_writer_equals_python_writer(tmp_path, workload, n_loci, monkeypatch):
    monkeypatch.setattr(make_synth_bam, "LOCI_PER_CONTIG", 300)  # several contigs in a small file
    monkeypatch.setattr(make_synth_bam, "CONTIG_LEN", 50_000 + 20_000 * 10_000 + 400_000)
    a, b = str(tmp_path / "py"), str(tmp_path / "native")
    n = make_synth_bam.write(workload, n_loci, a)
    assert make_synth_bam.write_native(workload, n_loci, b, threads=3) == n
    for ext in (".bam", ".bam.bai", ".bed"):
        assert filecmp.cmp(a + ext, b + ext, shallow=False), ext


@pytest.mark.parametrize("workload,n_loci,qual_mode", [("unphased100k", 40, 1), ("phased10k", 25, 0), ("expansion50k", 12, 1)])
def test_native_seq_writer_carries_the_same_alignments(tmp_path, workload, n_loci, qual_mode, monkeypatch, orc):
    """The SEQ / QUAL-bearing file (bench.py's l2_seq block) holds, record for record, the tid / pos / mapq / CIGAR / HP of the
    CIGAR-only file, SEQ and QUAL of the CIGAR's query length, HP as the last tag; records may span several BGZF blocks and
    several slabs; the rows through the host front end + oracle are those of the CIGAR-only file."""
    import gzip
    import struct

    import numpy as np

    from inquistr_amd import call, synth
    from tools import bamio

    monkeypatch.setattr(make_synth_bam, "LOCI_PER_CONTIG", 7)
    monkeypatch.setattr(make_synth_bam, "CONTIG_LEN", 50_000 + 20_000 * 10_000 + 400_000)
    a, b = str(tmp_path / "plain"), str(tmp_path / "seq")
    n = make_synth_bam.write_native(workload, n_loci, a, threads=3)
    info = {}
    assert make_synth_bam.write_native(workload, n_loci, b, threads=3, seq=True, qual_mode=qual_mode, slab_blocks=5, info=info) == n
    ua, ub = gzip.open(a + ".bam", "rb").read(), gzip.open(b + ".bam", "rb").read()
    assert len(ub) == info["inflated_bytes"] and info["n_blocks"] == (len(ub) + 0xFEFF) // 0xFF00

    def first_record(u):
        l_text = struct.unpack_from("<I", u, 4)[0]
        p = 8 + l_text
        (n_ref,) = struct.unpack_from("<I", u, p)
        p += 4
        for _ in range(n_ref):
            (l_name,) = struct.unpack_from("<I", u, p)
            p += 8 + l_name
        return p

    ra, rb = list(bamio.read_records(ua, first_record(ua))), list(bamio.read_records(ub, first_record(ub)))
    assert len(ra) == len(rb) == n and rb[-1]["next"] == len(ub)
    for x, y in zip(ra, rb):
        assert (x["tid"], x["pos"], x["mapq"], x["flag"], x["cigar"], x["hp"]) == (y["tid"], y["pos"], y["mapq"], y["flag"], y["cigar"], y["hp"])
        b0 = y["off"] + 4
        l_rn, n_cig, l_seq = ub[b0 + 8], struct.unpack_from("<H", ub, b0 + 12)[0], struct.unpack_from("<I", ub, b0 + 16)[0]
        assert l_seq == sum(w >> 4 for w in y["cigar"] if (w & 15) in (0, 1, 4, 7, 8)) and l_seq > 0
        qual = ub[b0 + 32 + l_rn + 4 * n_cig + (l_seq + 1) // 2 : b0 + 32 + l_rn + 4 * n_cig + (l_seq + 1) // 2 + l_seq]
        assert max(qual) <= 50
        assert ub[y["next"] - 4 : y["next"] - 1] == b"HPC"  # the last tag
    wl = synth.WORKLOADS[workload]
    rows = []
    for prefix in (a, b):
        fe = call.FrontEnd(prefix + ".bam", region_file=prefix + ".bed", minlen=wl.minlen, support=wl.support, unphased=wl.unphased, threads=2)
        g1, g2 = np.full(n_loci, -1.0), np.full(n_loci, -1.0)
        for batch, idx in fe.batches():
            code, res = orc.call_batch(batch)
            assert code == 0
            g1[idx], g2[idx] = res.phase1, res.phase2
        fe.close()
        rows.append((g1, g2))
    from tests import gen

    assert gen.same_f64(rows[0][0], rows[1][0]) and gen.same_f64(rows[0][1], rows[1][1])
    assert not np.all(np.isnan(rows[0][0]))
