"""BAM + BED -> .inq text through libinquistr_host.so + the HIP library, against text built from
the naive Python restatement.  Mirrors the reference's own tests (src/call.rs:525-582:
test_region, test_region_bed, test_unphased), which only smoke the path; here the text is checked
byte for byte."""
import functools
import os
import subprocess
import time

import numpy as np
import pytest

from inquistr_amd import call
from oracle import pyoracle as py
from tests.test_host_frontend import _expected, _make_case

pytestmark = pytest.mark.gpu


def _expected_text(loci, recs, unphased, minlen, support, sample, threads):
    p1, p2 = _expected(loci, recs, unphased, minlen, support)
    rows = [(c, s, e, a, b) for (c, s, e, _), a, b in zip(loci, p1, p2)]
    if threads > 1:  # src/call.rs:141 sort by (human chrom, start); equal keys keep BED order here
        rows.sort(key=functools.cmp_to_key(lambda x, y: py.human_compare(x[0], y[0]) or (x[1] > y[1]) - (x[1] < y[1])))
    return "\n".join([py.format_header(sample)] + [py.format_row(*r) for r in rows]) + "\n"


@pytest.mark.parametrize("seed,unphased,threads", [(11, False, 1), (12, True, 1), (13, False, 4), (14, True, 8)])
def test_genotype_repeats_text(tmp_path, seed, unphased, threads):
    minlen, support = 5, 3
    bam, bed, loci, recs = _make_case(tmp_path, seed, n_loci=80, ultra_long=(seed == 13))
    out = tmp_path / "out.inq"
    with open(out, "w") as f:
        call.genotype_repeats(bam, None, bed, minlen, support, threads, unphased, None, None, out=f)
    want = _expected_text(loci, recs, unphased, minlen, support, f"case{seed}.sorted", threads)
    assert open(out).read() == want


def test_cli_matches_library(tmp_path):
    bam, bed, loci, recs = _make_case(tmp_path, 21, n_loci=30)
    r = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "--sample-name", "sample", "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _expected_text(loci, recs, False, 5, 3, "sample", 4)
    # region string, unphased, custom minlen/support (src/call.rs:525-538 test_region shape)
    c, s, e, t = loci[0]
    r = subprocess.run([call.CLI_PATH, "call", bam, "-r", f"{c}:{s}-{e}", "-u", "-m", "3", "-s", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _expected_text([loci[0]], recs, True, 3, 2, "case21.sorted", 1)


def test_reference_bed_fixture_on_plumbing_bam(tmp_path):
    """BASELINE config #1 substitute: the reference's test-data/test.bed against a generated BAM whose
    header carries chr7 LN:159345973 (the only number the reference's tests assert, src/call.rs:604)."""
    import random

    from tests import gen
    from tools import bamio

    rng = random.Random(5)
    start, end = 154778571, 154779363
    recs = gen.random_locus_reads(rng, start, end, 40, long_every=6)
    recs.sort(key=lambda r: r.pos)
    bam = str(tmp_path / "plumbing.bam")
    w = bamio.BamWriter(bam, [("chr7", 159345973)])
    for i, r in enumerate(recs):
        tags = ([("HP", r.hp[0], r.hp[1])] if r.hp else []) + ([("SA", "Z", r.sa[1])] if r.sa else [])
        w.add(f"r{i}", r.flag, 0, r.pos, r.mapq, r.cigar, tags)
    w.close()
    bed = os.path.join(os.path.dirname(__file__), "golden", "reference_test.bed")
    for unphased in (False, True):
        out = tmp_path / f"o{unphased}.inq"
        with open(out, "w") as f:
            call.genotype_repeats(bam, None, bed, 5, 3, 4, unphased, "sample", None, out=f)
        want = _expected_text([("chr7", start, end, 0)], {0: recs}, unphased, 5, 3, "sample", 4)
        assert open(out).read() == want


@pytest.mark.parametrize("frontend", ["device", "host"])
def test_one_process_per_gpu_rehearsal(tmp_path, frontend):
    """`python -m torch.distributed.run ... -m inquistr_amd.call_dist` with two ranks (gloo, both on this box's one
    GPU): every rank runs the C++ driver on its slice, rank 0 gathers; the .inq equals the single-process CLI's."""
    import subprocess
    import sys

    from inquistr_amd import call
    from tools import make_synth_bam

    prefix = str(tmp_path / "w")
    make_synth_bam.write_native("phased10k", 4000, prefix)
    single = tmp_path / "single.inq"
    with open(single, "w") as f:
        call.genotype_repeats(prefix + ".bam", None, prefix + ".bed", 5, 3, 4, False, "S", None, out=f, frontend=frontend)
    out = tmp_path / "dist.inq"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", "-m", "inquistr_amd.call_dist", prefix + ".bam", "-R", prefix + ".bed", "-t", "4",
           "--sample-name", "S", "--backend", "gloo", "--same-device", "--frontend", frontend, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == single.read_text() and out.read_text().count("\n") == 4001


@pytest.mark.parametrize("frontend", ["device", "host", None])
def test_session_many_bams_equal_separate_calls(tmp_path, frontend):
    """inq_session_call_many (one HIP context for a cohort; file k + 1 staged while file k is called) against one
    inq_genotype_repeats per BAM: byte for byte the same .inq per file, a failing file in the middle does not disturb the
    others, and `combine` of the session's outputs equals `combine` of the separate ones (src/combine.rs:27-59)."""
    from tools import make_synth_bam

    names, bed = [], None
    for k, (wl, n) in enumerate([("phased10k", 700), ("phased10k", 350), ("expansion50k", 90), ("phased10k", 1200)]):
        prefix = str(tmp_path / f"s{k}")
        make_synth_bam.write_native(wl, n, prefix, threads=4)
        names.append(prefix)
    beds = [p + ".bed" for p in names]
    missing = str(tmp_path / "missing.bam")
    order = [names[0] + ".bam", names[1] + ".bam", missing, names[2] + ".bam", names[3] + ".bam"]
    # every file against the SHORTEST bed (loci 0 .. 89 exist in all four files: same coordinates, generator positions)
    bed = beds[2]
    sep = []
    for b in order:
        if b == missing:
            sep.append(None)
            continue
        o = tmp_path / (os.path.basename(b) + ".sep.inq")
        with open(o, "w") as f:
            call.genotype_repeats(b, None, bed, 5, 3, 4, False, None, None, out=f, frontend=frontend)
        sep.append(o.read_text())
    outs = [open(tmp_path / f"sess{k}.inq", "w") for k in range(len(order))]
    with call.Session(0) as S:
        st = S.call_many(order, outs, region_file=bed, threads=4, frontend=frontend)
        assert st == [0, 0, 1, 0, 0], (st, S.last_message)
        assert "missing.bam" in S.last_message
        # the session stays usable: one more call on the same context
        again = tmp_path / "again.inq"
        with open(again, "w") as f:
            S.call(order[0], region_file=bed, threads=4, out=f, frontend=frontend)
    for o in outs:
        o.close()
    for k, b in enumerate(order):
        if sep[k] is not None:
            assert (tmp_path / f"sess{k}.inq").read_text() == sep[k], b
    assert again.read_text() == sep[0]
    # combine of the cohort
    good = [k for k in range(len(order)) if sep[k] is not None]
    a, b = tmp_path / "comb_sess.inq", tmp_path / "comb_sep.inq"
    for dst, files in ((a, [str(tmp_path / f"sess{k}.inq") for k in good]),
                       (b, [str(tmp_path / (os.path.basename(order[k]) + ".sep.inq")) for k in good])):
        with open(dst, "w") as f:
            call.combine(files, out=f)
    assert a.read_text() == b.read_text() and a.read_text().count("\n") == 91


def test_cohort_command_equals_call_per_bam(tmp_path):
    """`inquistr cohort` (many `call`s in one process) writes, per BAM, what `inquistr call` prints, and --combined what
    `inquistr combine` of those prints."""
    from tools import make_synth_bam

    prefixes = []
    for k in range(3):
        prefix = str(tmp_path / f"c{k}")
        make_synth_bam.write_native("unphased100k", 500 + 100 * k, prefix, threads=4)
        prefixes.append(prefix)
    bed = prefixes[0] + ".bed"
    outdir = tmp_path / "out"
    outdir.mkdir()
    r = subprocess.run([call.CLI_PATH, "cohort", "-R", bed, "-u", "-t", "4", "--out-dir", str(outdir), "--combined", str(tmp_path / "all.inq")]
                       + [p + ".bam" for p in prefixes], capture_output=True, text=True, env=dict(os.environ, INQ_FRONTEND="device"))
    assert r.returncode == 0, r.stderr
    texts = []
    for k, p in enumerate(prefixes):
        one = subprocess.run([call.CLI_PATH, "call", p + ".bam", "-R", bed, "-u", "-t", "4"], capture_output=True, text=True,
                             env=dict(os.environ, INQ_FRONTEND="device"))
        assert one.returncode == 0, one.stderr
        assert (outdir / f"c{k}.inq").read_text() == one.stdout
        (tmp_path / f"one{k}.inq").write_text(one.stdout)
        texts.append(str(tmp_path / f"one{k}.inq"))
    comb = subprocess.run([call.CLI_PATH, "combine"] + texts, capture_output=True, text=True)
    assert comb.returncode == 0 and (tmp_path / "all.inq").read_text() == comb.stdout


@pytest.mark.parametrize("frontend", ["device", "host"])
def test_session_survives_a_file_that_fails_on_the_device(tmp_path, frontend):
    """A file whose failure is only seen on the GPU - a record the reference panics on (HP typed `s`, phased mode: get_phase,
    src/call.rs:482-491), behind spans that were already appended to the device-resident batch - ends with status 101; the files
    behind it on the same context are not disturbed (nothing of the failed file's batch stays behind)."""
    from tools import bamio

    def write(path, bad_at=None, n=60):
        w = bamio.BamWriter(path, [("chr1", 3_000_000)], block=3000)
        for k in range(n):
            pos = 10_000 + 20_000 * k
            for r in range(8):
                hp = ("s", 1) if (bad_at == k and r == 3) else ("C", 1 + r % 2)
                w.add(f"r{k}_{r}", 0, 0, pos - 300, 60, [("M", 400), ("I", 9 + k % 5), ("M", 400)], [("HP", hp[0], hp[1])])
        w.close()

    good1, bad, good2 = str(tmp_path / "g1.bam"), str(tmp_path / "bad.bam"), str(tmp_path / "g2.bam")
    write(good1)
    write(bad, bad_at=45)
    write(good2, n=50)
    bed = str(tmp_path / "loci.bed")
    with open(bed, "w") as f:
        f.write("".join(f"chr1\t{10_000 + 20_000 * k}\t{10_050 + 20_000 * k}\n" for k in range(60)))
    env_before = dict(os.environ)
    os.environ["INQ_SPAN_MB"] = "0"
    os.environ["INQ_SPAN_GAP_BYTES"] = "0"  # every locus its own segment; with INQ_FLUSH_LOCI the failing span comes behind a flush
    os.environ["INQ_FLUSH_LOCI"] = "7"
    try:
        want = {}
        for b in (good1, good2):
            o = tmp_path / (os.path.basename(b) + ".sep")
            with open(o, "w") as f:
                call.genotype_repeats(b, None, bed, 5, 3, 2, False, None, None, out=f, frontend=frontend)
            want[b] = o.read_text()
        outs = [open(tmp_path / f"o{k}.inq", "w") for k in range(4)]
        with call.Session(0) as S:
            st = S.call_many([good1, bad, good2, good1], outs, region_file=bed, threads=2, frontend=frontend)
        for o in outs:
            o.close()
        assert st == [0, 101, 0, 0], (st, S.last_message)
        assert (tmp_path / "o0.inq").read_text() == want[good1] and (tmp_path / "o3.inq").read_text() == want[good1]
        assert (tmp_path / "o2.inq").read_text() == want[good2]
        assert (tmp_path / "o1.inq").read_text() == ""  # the reference panics before it prints with -t >= 2 (rows are collected first)
    finally:
        os.environ.clear()
        os.environ.update(env_before)


def test_call_through_a_resident_server_equals_call(tmp_path):
    """`inquistr serve` keeps the device context; `inquistr call` with INQ_SERVER set hands it the arguments (relative paths made
    absolute) and its stdout: same bytes, same exit status and message as the call run by itself - also for a file the reference
    panics on, after which the server still answers -, and without a server at that address the call runs by itself."""
    from tools import make_synth_bam

    prefixes = []
    for k in range(3):
        prefix = str(tmp_path / f"s{k}")
        make_synth_bam.write_native("unphased100k" if k != 1 else "phased10k", 400 + 100 * k, prefix, threads=4)
        prefixes.append(prefix)
    sock = str(tmp_path / "inq.sock")
    env_direct = dict(os.environ, INQ_FRONTEND="device")
    env_direct.pop("INQ_SERVER", None)
    env_served = dict(env_direct, INQ_SERVER=sock)
    # no server yet: the call runs by itself
    alone = subprocess.run([call.CLI_PATH, "call", prefixes[0] + ".bam", "-R", prefixes[0] + ".bed", "-u", "-t", "4"], capture_output=True,
                           text=True, env=env_served)
    assert alone.returncode == 0 and alone.stdout.count("\n") == 401
    server = subprocess.Popen([call.CLI_PATH, "serve", "--socket", sock, "--idle-exit", "120"], env=env_direct, stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock)
        for rep in range(2):
            for k, p in enumerate(prefixes):
                flags = ["-t", "4"] + ([] if k == 1 else ["-u"])
                want = subprocess.run([call.CLI_PATH, "call", p + ".bam", "-R", p + ".bed"] + flags, capture_output=True, text=True, env=env_direct)
                got = subprocess.run([call.CLI_PATH, "call", os.path.basename(p) + ".bam", "-R", os.path.basename(p) + ".bed"] + flags,
                                     capture_output=True, text=True, env=env_served, cwd=str(tmp_path))
                assert want.returncode == 0 and got.returncode == 0, (want.stderr, got.stderr)
                assert got.stdout == want.stdout and got.stdout.count("\n") == 401 + 100 * k
        # errors travel: a missing file (exit 1), a locus on an unknown contig (the reference's panic, exit 101)
        for args in (["nope.bam", "-R", prefixes[0] + ".bed"], [prefixes[0] + ".bam", "-r", "chrNope:100-200"]):
            want = subprocess.run([call.CLI_PATH, "call"] + args, capture_output=True, text=True, env=env_direct, cwd=str(tmp_path))
            got = subprocess.run([call.CLI_PATH, "call"] + args, capture_output=True, text=True, env=env_served, cwd=str(tmp_path))
            assert got.returncode == want.returncode != 0 and got.stdout == want.stdout
            if "-r" in args:  # (the other message names the file, by the path each side was given)
                assert got.stderr.strip().splitlines()[-1] == want.stderr.strip().splitlines()[-1]
        again = subprocess.run([call.CLI_PATH, "call", prefixes[2] + ".bam", "-R", prefixes[2] + ".bed", "-u"], capture_output=True, text=True, env=env_served)
        assert again.returncode == 0 and again.stdout.count("\n") == 601
        # four callers at once (a workflow manager's way; four, so that even without the server the box's limit of six processes on the
        # card holds): they queue, file k + 1 is staged while file k is called, every one gets its own rows
        procs = []
        for k in (0, 1, 2, 1):
            flags = ["-t", "4"] + ([] if k == 1 else ["-u"])
            # (stdout into a file: rows of a caller whose pipe nobody reads yet would stall the server's queue)
            f = open(tmp_path / f"par{len(procs)}.inq", "w")
            procs.append((k, f, subprocess.Popen([call.CLI_PATH, "call", prefixes[k] + ".bam", "-R", prefixes[k] + ".bed"] + flags, stdout=f,
                                                 stderr=subprocess.PIPE, text=True, env=env_served)))
        wants = {}
        for k, f, pr in procs:
            _, err = pr.communicate(timeout=120)
            assert pr.returncode == 0, err
            f.close()
            out = open(f.name).read()
            if k not in wants:
                flags = ["-t", "4"] + ([] if k == 1 else ["-u"])
                wants[k] = subprocess.run([call.CLI_PATH, "call", prefixes[k] + ".bam", "-R", prefixes[k] + ".bed"] + flags, capture_output=True,
                                          text=True, env=env_direct).stdout
            assert out == wants[k]
        quit_ = subprocess.run([call.CLI_PATH, "serve", "--socket", sock, "--quit"], capture_output=True, text=True)
        assert quit_.returncode == 0
        assert server.wait(timeout=30) == 0
        assert "leaving after 13 calls" in server.stderr.read()
    finally:
        if server.poll() is None:
            server.kill()
            server.wait(timeout=30)


def test_session_target_list_is_reused_only_when_it_may_be(tmp_path):
    """A session keeps the parsed and validated BED of the file before (a cohort has one BED): it is taken again for the same
    file and the same contigs - and only then.  A BAM whose contig is shorter makes the reference panic on a locus beyond its
    end (src/repeats.rs:108-114) although the BAM before accepted it; a BED rewritten in place gives the new rows."""
    from tools import bamio

    def write(path, ln):
        w = bamio.BamWriter(path, [("chr1", ln)], block=3000)
        for k in range(30):
            pos = 10_000 + 20_000 * k
            for r in range(8):
                w.add(f"r{k}_{r}", 0, 0, pos - 300, 60, [("M", 400), ("I", 9 + k % 5), ("M", 400)], [("HP", "C", 1 + r % 2)])
        w.close()

    long_, short = str(tmp_path / "long.bam"), str(tmp_path / "short.bam")
    write(long_, 3_000_000)
    write(short, 650_000)
    bed = str(tmp_path / "loci.bed")

    def put_bed(ks):
        with open(bed, "w") as f:
            f.write("".join(f"chr1\t{10_000 + 20_000 * k + 90}\t{10_000 + 20_000 * k + 110}\n" for k in ks))

    def direct(bam):
        r = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "-t", "2"], capture_output=True, text=True, env=dict(os.environ, INQ_FRONTEND="device"))
        return r.returncode, r.stdout

    def in_session(S, bam, name):
        o = tmp_path / name
        try:
            with open(o, "w") as f:
                S.call(bam, region_file=bed, threads=2, out=f, frontend="device")
            return 0, o.read_text()
        except call.CallError as e:
            return e.status, o.read_text()

    put_bed(range(0, 29))  # the last locus ends at 570 110: inside both contigs
    with call.Session(0) as S:
        want_long, want_short = direct(long_), direct(short)
        assert want_long[0] == want_short[0] == 0 and want_long[1].count("\n") == 30
        assert in_session(S, long_, "a.inq") == want_long
        assert in_session(S, long_, "b.inq") == want_long      # the kept list
        assert in_session(S, short, "c.inq") == want_short     # same BED, other contig length: validated again, still fine
        put_bed(range(0, 30))  # + a locus at 590 090 .. 590 110 ... and one beyond the short contig's end
        with open(bed, "a") as f:
            f.write("chr1\t700000\t700050\n")
        st = os.stat(bed)
        os.utime(bed, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
        want_long2, want_short2 = direct(long_), direct(short)
        assert want_long2[0] == 0 and want_long2[1].count("\n") == 32 and want_short2[0] == 101
        assert in_session(S, long_, "d.inq") == want_long2     # the rewritten BED is read again
        assert in_session(S, short, "e.inq")[0] == 101         # the list the long contig accepted is not taken for the short one
        assert in_session(S, long_, "f.inq") == want_long2


def test_auto_server_gives_the_calls_own_rows(tmp_path):
    """INQ_SERVER=auto: the first `inquistr call` starts the user's server for the device and is served by it, the next ones find
    it; rows, status and message are the call's own.  The server is asked to leave at the end (it would by itself when idle)."""
    from tools import make_synth_bam

    prefix = str(tmp_path / "a")
    make_synth_bam.write_native("unphased100k", 700, prefix, threads=4)
    env_direct = dict(os.environ, INQ_FRONTEND="device")
    env_direct.pop("INQ_SERVER", None)
    env_auto = dict(env_direct, INQ_SERVER="auto", XDG_RUNTIME_DIR=str(tmp_path), INQ_SERVER_IDLE="30")
    sock = tmp_path / f"inquistr-{os.getuid()}-dev0.sock"
    cmd = [call.CLI_PATH, "call", prefix + ".bam", "-R", prefix + ".bed", "-u", "-t", "4"]
    want = subprocess.run(cmd, capture_output=True, text=True, env=env_direct)
    assert want.returncode == 0 and want.stdout.count("\n") == 701
    try:
        for _ in range(3):
            got = subprocess.run(cmd, capture_output=True, text=True, env=env_auto, timeout=120)
            assert got.returncode == 0 and got.stdout == want.stdout and sock.exists()
        bad = subprocess.run([call.CLI_PATH, "call", prefix + ".bam", "-r", "chrNope:100-200"], capture_output=True, text=True, env=env_auto, timeout=120)
        assert bad.returncode == 101 and bad.stdout == ""
    finally:
        q = subprocess.run([call.CLI_PATH, "serve", "--socket", str(sock), "--quit"], capture_output=True, text=True, timeout=60)
    assert q.returncode == 0


@pytest.mark.parametrize("workload,n_loci,level", [("unphased100k", 60, 6), ("phased10k", 45, 6), ("expansion50k", 14, 1)])
def test_native_seq_bearing_file_through_the_device_front_end(tmp_path, monkeypatch, workload, n_loci, level):
    """Records shaped like a long-read BAM (SEQ + QUAL of the query length, NM, ML:B,C + MM:Z, HP as the LAST tag; ~18 KB each, most
    of them cut by BGZF block ends) written by the native writer bench.py's l2_seq blocks use (inq_synth_write_bam_seq), at zlib
    level 6 and 1: the device front end - inflate, CRC, record chain, aux walk over the long tags, gather, join, locus kernels,
    several spans - must print the rows the plain-Python restatement computes from the records a plain-Python reader finds in the
    same file (tools/bamio.py on gzip: no code shared with the product)."""
    import gzip
    import struct

    from inquistr_amd import synth
    from tools import bamio, make_synth_bam

    monkeypatch.setattr(make_synth_bam, "LOCI_PER_CONTIG", 9)
    monkeypatch.setattr(make_synth_bam, "CONTIG_LEN", 50_000 + 20_000 * 10_000 + 400_000)
    prefix = str(tmp_path / "seq")
    info = {}
    n = make_synth_bam.write_native(workload, n_loci, prefix, threads=3, seq=True, level=level, slab_blocks=7, info=info)
    wl = synth.WORKLOADS[workload]
    u = gzip.open(prefix + ".bam", "rb").read()
    assert len(u) == info["inflated_bytes"] and len(u) > 12_000 * n  # the records really carry their bases and qualities
    l_text = struct.unpack_from("<I", u, 4)[0]
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<I", u, p)
    p += 4
    names = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<I", u, p)
        names.append(u[p + 4 : p + 4 + l_name - 1].decode())
        p += 8 + l_name
    recs = {t: [] for t in range(n_ref)}
    n_seen = 0
    for r in bamio.read_records(u, p):
        n_seen += 1
        recs[r["tid"]].append(py.Record(pos=r["pos"], cigar=[(py.OPS[w & 15], w >> 4) for w in r["cigar"]], mapq=r["mapq"], flag=r["flag"],
                                        tid=r["tid"], hp=r["hp"], sa=r["sa"]))
    assert n_seen == n
    loci = []
    for ln in open(prefix + ".bed"):
        c, s, e = ln.split()
        loci.append((c, int(s), int(e), names.index(c)))
    assert len(loci) == n_loci
    want = _expected_text(loci, recs, wl.unphased, wl.minlen, wl.support, "S", 4)
    cmd = [call.CLI_PATH, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", "4", "--sample-name", "S", "-m", str(wl.minlen), "-s", str(wl.support)]
    cmd += ["-u"] if wl.unphased else []
    for span_mb in ("256", "1"):  # one span, and several (a 1 MB span holds a few dozen of these records)
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, INQ_FRONTEND="device", INQ_SPAN_MB=span_mb, INQ_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert r.stdout == want, f"rows differ with {span_mb} MB spans"
        if span_mb == "1":
            import re

            m = re.search(r"span loop: (\d+) spans", r.stderr)
            assert m and int(m.group(1)) > 1
    host = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, INQ_FRONTEND="host"))
    assert host.returncode == 0 and host.stdout == want


def test_four_ranks_on_one_gpu_share_the_bytes_and_print_the_same_rows(tmp_path, monkeypatch):
    """`call_dist` at world size 4, all ranks on this box's one GPU (gloo for the gather; with pytest's own process five processes
    on the card), over a multi-contig file of long-read-shaped records (SEQ, QUAL, ML / MM, HP last): inq_run_partition cuts the
    targets so that every rank reads about the same number of BAM bytes - each rank reports what its device front end was handed
    and the four figures must lie within 15 % of their mean - and the gathered .inq equals the single-process CLI's, byte for byte."""
    import json
    import sys

    from inquistr_amd import synth
    from tools import make_synth_bam

    monkeypatch.setattr(make_synth_bam, "LOCI_PER_CONTIG", 700)
    monkeypatch.setattr(make_synth_bam, "CONTIG_LEN", 50_000 + 20_000 * 10_000 + 400_000)  # (a locus' position comes from its number in the workload)
    prefix = str(tmp_path / "w")
    n_loci = 3_000
    make_synth_bam.write_native("unphased100k", n_loci, prefix, seq=True, level=6)
    wl = synth.WORKLOADS["unphased100k"]
    assert os.path.getsize(prefix + ".bam") > 600e6
    single = subprocess.run([call.CLI_PATH, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", "4", "--sample-name", "S", "-u"],
                            capture_output=True, text=True, env=dict(os.environ, INQ_FRONTEND="device"))
    assert single.returncode == 0, single.stderr
    assert single.stdout.count("\n") == n_loci + 1 and len({ln.split("\t")[0] for ln in single.stdout.splitlines()[1:]}) == 5  # five contigs
    out, sdir = tmp_path / "dist.inq", tmp_path / "stats"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29641", "-m", "inquistr_amd.call_dist", prefix + ".bam", "-R", prefix + ".bed", "-t", "4", "-u",
           "--sample-name", "S", "--backend", "gloo", "--same-device", "--frontend", "device", "-o", str(out), "--stats-dir", str(sdir)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == single.stdout
    st = [json.load(open(sdir / f"rank{k}.json")) for k in range(4)]
    assert sum(s["loci"] for s in st) == n_loci and all(s["world"] == 4 for s in st)
    read = [s["bam_bytes_read"] for s in st]
    mean = sum(read) / 4
    assert mean > 100e6, read  # every rank went through the device front end
    assert all(abs(b - mean) <= 0.15 * mean for b in read), f"BAM bytes per rank {read} (mean {mean:.0f})"
    # what the ranks read together is the file once, plus what neighbours both need at the three cuts
    assert sum(read) < 1.1 * os.path.getsize(prefix + ".bam")


@pytest.mark.parametrize("workload,n_loci,seq", [("unphased100k", 20_000, False), ("phased10k", 10_000, False), ("expansion50k", 1_500, True)])
def test_file_scale_rows_equal_the_cpu_program_whatever_the_span_size(tmp_path, workload, n_loci, seq):
    """At file scale (BASELINE's configs #3, #2 and #5 cut to 10 - 20 000 loci, zlib level 6; config #5 with SEQ / QUAL-bearing
    records): the `.inq` of the product CLI is the same for 256 MB, 16 MB and 3 MB spans (how the file is cut, how many spans are in
    flight, how often the batch is flushed must not show), the same through the host front end, and byte for byte what
    oracle/ref_shaped_call prints - the reference's control flow on the CPU oracle with its OWN BGZF / BAM / BAI reader (zlib), no
    code shared with the product."""
    from inquistr_amd import synth
    from tools import make_synth_bam

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "ref_shaped_call"], stdout=subprocess.DEVNULL)
    ref = os.path.join(root, "oracle", "ref_shaped_call")
    wl = synth.WORKLOADS[workload]
    prefix = str(tmp_path / "f")
    make_synth_bam.write_native(workload, n_loci, prefix, level=6, seq=seq)
    un = ["-u"] if wl.unphased else []
    cmd = [call.CLI_PATH, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", "8", "--sample-name", "S", "-m", str(wl.minlen), "-s", str(wl.support)] + un
    want = subprocess.run([ref, prefix + ".bam", prefix + ".bed", "B", "8", str(int(wl.unphased)), str(wl.minlen), str(wl.support), "S"],
                          capture_output=True)
    assert want.returncode == 0, want.stderr[-500:]
    assert want.stdout.count(b"\n") == n_loci + 1
    for env, extra in (({"INQ_FRONTEND": "device"}, []), ({"INQ_FRONTEND": "device", "INQ_SPAN_MB": "16", "INQ_FLUSH_LOCI": "3000"}, []),
                       ({"INQ_FRONTEND": "device", "INQ_SPAN_MB": "3"}, ["--ctx-option", "inflate_ahead=0", "--ctx-option=gather_nt=0"]),
                       ({"INQ_FRONTEND": "host"}, [])):
        r = subprocess.run(cmd + extra, capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-500:]
        assert r.stdout == want.stdout, f"rows differ with {env} {extra}"
    bad = subprocess.run(cmd + ["--ctx-option", "inflate_ahead"], capture_output=True)
    assert bad.returncode == 2 and b"--ctx-option" in bad.stderr
    bad = subprocess.run(cmd + ["--ctx-option", "no_such=1"], capture_output=True)
    assert bad.returncode == 2 and b"no such option" in bad.stderr


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_native_devices_entry_on_one_gpu_equals_the_single_device_call(tmp_path, monkeypatch, devices):
    """`inquistr call --devices 0,0[,0,0]` (inq_genotype_repeats_devices: ONE process, one thread + one device context per listed
    device, no torch, no collective) over the multi-contig long-read-shaped file: the .inq equals the single-device CLI's byte for
    byte, every part went through the device front end on about the same number of BAM bytes, and the parts' reader pools share the
    granted cores."""
    import re

    from tools import make_synth_bam

    monkeypatch.setattr(make_synth_bam, "LOCI_PER_CONTIG", 700)
    monkeypatch.setattr(make_synth_bam, "CONTIG_LEN", 50_000 + 20_000 * 10_000 + 400_000)
    prefix = str(tmp_path / "w")
    n_loci = 3_000
    make_synth_bam.write_native("unphased100k", n_loci, prefix, seq=True, level=6)
    base = [call.CLI_PATH, "call", prefix + ".bam", "-R", prefix + ".bed", "-t", "4", "--sample-name", "S", "-u"]
    env = dict(os.environ, INQ_FRONTEND="device", INQ_TIMING="1")
    single = subprocess.run(base, capture_output=True, text=True, env=env)
    assert single.returncode == 0, single.stderr
    multi = subprocess.run(base + ["--devices", devices], capture_output=True, text=True, env=env, timeout=600)
    assert multi.returncode == 0, multi.stderr[-2000:]
    assert multi.stdout == single.stdout and multi.stdout.count("\n") == n_loci + 1
    n = devices.count(",") + 1
    parts = re.findall(r"\[inq part\] (\d+) of (\d+) on device 0: status 0, (\d+) loci, (\d+) spans, ([\d.]+) MB of BAM read by (\d+) reader threads", multi.stderr)
    assert len(parts) == n and all(int(p[1]) == n for p in parts)
    assert sum(int(p[2]) for p in parts) == n_loci
    read = [float(p[4]) for p in parts]
    mean = sum(read) / n
    assert mean > 100 and all(abs(b - mean) <= 0.15 * mean for b in read), read
    assert sum(read) * 1e6 < 1.1 * os.path.getsize(prefix + ".bam")
    granted = call.load().inq_host_granted_cpus()
    assert all(int(p[5]) == min(8, max(2, granted // n)) for p in parts), (parts, granted)


def test_native_devices_entry_phased(tmp_path):
    """Phased mode (the reference's default, src/main.rs:55) through the library entry with three parts on the one GPU - a small
    multi-contig file with nested loci, an ultra-long read, contigs without reads - against the single-device call and the
    plain-Python restatement."""
    bam, bed, loci, recs = _make_case(tmp_path, 77, n_loci=60, ultra_long=True)
    single = tmp_path / "single.inq"
    with open(single, "w") as f:
        call.genotype_repeats(bam, None, bed, 5, 3, 4, False, "S", out=f, frontend="device")
    multi = tmp_path / "multi.inq"
    with open(multi, "w") as f:
        st = call.genotype_repeats_devices(bam, None, bed, [0, 0, 0], threads=4, sample_name="S", out=f, frontend="device")
    assert multi.read_text() == single.read_text() == _expected_text(loci, recs, False, 5, 3, "S", 4)
    assert len(st) == 3 and sum(s["loci"] for s in st) == len(loci) and all(s["status"] == 0 and s["front"] == 2 for s in st)


def test_ctx_create_multi_on_one_gpu():
    """inq_ctx_create_multi (SURVEY 8b's list form): three contexts on the one device, made concurrently, each usable on its own."""
    from inquistr_amd import hipcall, synth

    ctxs = hipcall.Context.create_multi([0, 0, 0])
    try:
        assert len(ctxs) == 3 and all("gfx950" in c.backend for c in ctxs)
        b = synth.generate_numpy(synth.WORKLOADS["phased10k"], 0, 64)
        rows = [c.call_batch(b)[1] for c in ctxs]
        for r in rows[1:]:
            assert np.array_equal(np.nan_to_num(r.phase1, nan=-1e300), np.nan_to_num(rows[0].phase1, nan=-1e300))
    finally:
        for c in ctxs:
            c.close()
    with pytest.raises(hipcall.InqError):
        hipcall.Context.create_multi([0, 99])  # all or nothing: the good context is taken back


@pytest.mark.parametrize("frontend,unphased", [("device", False), ("device", True), ("host", False)])
def test_rows_left_in_device_memory_equal_the_host_rows(tmp_path, frontend, unphased):
    """inq_run_rows_device (what the RCCL gather of inquistr_amd/call_dist.py reads from): the rows of a share of the targets left in
    a [2][width] device buffer - written by the flushes' scatter (device front end) or copied up once (host sweep) -, NaN behind the
    share and for loci no span holds, against inq_run_rows on the same share; then call_dist with rows='device' at world size 1
    against the CLI's text."""
    import torch

    from inquistr_amd import call_dist

    bam, bed, loci, recs = _make_case(tmp_path, 78, n_loci=90, ultra_long=True)
    run = call.Run(bam, None, bed, 5, 3, 4, unphased, "S", frontend=frontend)
    order, cuts = run.partition(3)
    mine = order[int(cuts[1]):int(cuts[2])]
    width = len(mine) + 7
    want1, want2 = run.rows(mine)
    d1, d2 = run.rows_device(mine, width)
    assert d2 == d1 + 8 * width
    t = torch.as_tensor(call_dist._DeviceArray(d1, (2, width)), device="cuda:0").cpu().numpy()
    assert np.array_equal(np.nan_to_num(t[0, : len(mine)], nan=-7e77), np.nan_to_num(want1, nan=-7e77))
    assert np.array_equal(np.nan_to_num(t[1, : len(mine)], nan=-7e77), np.nan_to_num(want2, nan=-7e77))
    assert np.isnan(t[:, len(mine):]).all() and (~np.isnan(want1)).sum() > 3
    run.close()
    out = tmp_path / "d.inq"
    with open(out, "w") as f:
        call_dist.genotype_repeats_distributed(bam, None, bed, 5, 3, 4, unphased, "S", out=f, frontend=frontend, rows="device")
    assert out.read_text() == _expected_text(loci, recs, unphased, 5, 3, "S", 4)


@pytest.mark.parametrize("frontend", ["device", "host"])
def test_blocking_waits_print_the_same_rows(tmp_path, frontend):
    """"blocking_sync" (every wait of the context gives its core back: what the host library picks by itself for a caller whose share
    of the granted cores is below 8, here forced either way through the CLI's --ctx-option, and picked by the library for a process
    confined to two CPUs): the same text as with spinning waits, on both front ends."""
    bam, bed, loci, recs = _make_case(tmp_path, 61, n_loci=70, ultra_long=True)
    want = _expected_text(loci, recs, False, 5, 3, "S", 4)
    env = dict(os.environ, INQ_FRONTEND=frontend)
    for extra in (["--ctx-option", "blocking_sync=1"], ["--ctx-option", "blocking_sync=0"]):
        r = subprocess.run([call.CLI_PATH, "call", bam, "-R", bed, "--sample-name", "S", "-t", "4"] + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout == want, extra
    cpus = sorted(os.sched_getaffinity(0))[:2]
    r = subprocess.run(["taskset", "-c", ",".join(map(str, cpus)), call.CLI_PATH, "call", bam, "-R", bed, "--sample-name", "S", "-t", "4"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want


@pytest.mark.parametrize("frontend", ["device", "host"])
def test_runs_on_one_session_equal_runs_of_their_own(tmp_path, frontend):
    """inq_session_run_open: runs opened on ONE session (its device context, span buffers and BED cache - what a resident rank of
    call_dist keeps for the life of the process) give the rows of runs that make their own context, file after file, host rows and
    device rows alike; and call_dist, which takes the process's session by itself, prints the same text pass after pass."""
    import torch

    from inquistr_amd import call_dist

    cases = [_make_case(tmp_path / f"c{k}", 90 + k, n_loci=60 + 11 * k, ultra_long=True) for k in range(2) if not (tmp_path / f"c{k}").mkdir()]
    with call.Session(0) as S:
        for rep in range(2):
            for bam, bed, loci, recs in cases:
                for unphased in (False, True):
                    own = call.Run(bam, None, bed, 5, 3, 4, unphased, "S", frontend=frontend)
                    order, cuts = own.partition(2)
                    mine = order[int(cuts[1]):]
                    want1, want2 = own.rows(mine)
                    own.close()
                    srun = call.Run(bam, None, bed, 5, 3, 4, unphased, "S", frontend=frontend, session=S)
                    got1, got2 = srun.rows(mine)
                    assert np.array_equal(np.nan_to_num(got1, nan=-7e77), np.nan_to_num(want1, nan=-7e77))
                    assert np.array_equal(np.nan_to_num(got2, nan=-7e77), np.nan_to_num(want2, nan=-7e77))
                    d1, _ = srun.rows_device(mine, len(mine) + 3)
                    t = torch.as_tensor(call_dist._DeviceArray(d1, (2, len(mine) + 3)), device="cuda:0").cpu().numpy()
                    assert np.array_equal(np.nan_to_num(t[0, : len(mine)], nan=-7e77), np.nan_to_num(want1, nan=-7e77))
                    assert np.array_equal(np.nan_to_num(t[1, : len(mine)], nan=-7e77), np.nan_to_num(want2, nan=-7e77))
                    assert (~np.isnan(want1)).sum() > 1
                    srun.close()
    for rep in range(3):
        bam, bed, loci, recs = cases[rep % 2]
        out = tmp_path / f"p{rep}.inq"
        st = {}
        with open(out, "w") as f:
            call_dist.genotype_repeats_distributed(bam, None, bed, 5, 3, 4, rep == 1, "S", out=f, frontend=frontend, stats=st)
        assert out.read_text() == _expected_text(loci, recs, rep == 1, 5, 3, "S", 4)
        assert st["open_s"] > 0 and st["rows_s"] > 0
    assert len(call_dist._sessions) == 1  # one context for the three passes


def test_rows_gathered_over_rccl_from_device_memory(tmp_path):
    """The collective of the one-process-per-GPU form on what a one-GPU box offers: a `nccl` (= RCCL) process group of world size 1,
    rows left in device memory by the flushes (rows='device'), `dist.gather` reading them there, rank 0 copying the gathered block
    down and writing the text - against the CLI-equivalent text of the plain-Python restatement."""
    import socket

    import torch
    import torch.distributed as dist

    from inquistr_amd import call_dist

    bam, bed, loci, recs = _make_case(tmp_path, 79, n_loci=70, ultra_long=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        for unphased in (False, True):
            out = tmp_path / f"g{int(unphased)}.inq"
            st = {}
            with open(out, "w") as f:
                call_dist.genotype_repeats_distributed(bam, None, bed, 5, 3, 4, unphased, "S", out=f, frontend="device", rows="device", stats=st)
            assert out.read_text() == _expected_text(loci, recs, unphased, 5, 3, "S", 4)
            assert st["rows_in"] == "device memory" and st["gather_s"] >= 0
    finally:
        dist.destroy_process_group()
