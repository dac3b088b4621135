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
