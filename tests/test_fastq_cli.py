"""FASTQ reader (CPU) and the two CLI entry points end to end (GPU)."""
import gzip
import os
import tempfile

import numpy as np
import pytest

import fixtures as fx
from metamlst_amd import synth
from metamlst_amd.fastq import interleave, read_batches


def write_fastq(path, bases, quals, gz=False):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for k in range(bases.shape[0]):
            f.write(b"@r%d extra\n" % k + bases[k].tobytes() + b"\n+\n" + quals[k].tobytes() + b"\n")


def test_fastq_roundtrip_plain_gz_and_batches():
    rng = np.random.default_rng(0)
    b = synth._ACGT[rng.integers(0, 4, size=(25, 60))]
    q = (rng.integers(2, 41, size=(25, 60)) + 33).astype(np.uint8)
    d = tempfile.mkdtemp()
    write_fastq(d + "/a.fastq", b, q)
    write_fastq(d + "/a.fastq.gz", b, q, gz=True)
    for path in (d + "/a.fastq", d + "/a.fastq.gz"):
        got = list(read_batches(path, batch_reads=10))
        assert [len(g[2]) - 1 for g in got] == [10, 10, 5]
        assert np.array_equal(np.concatenate([g[0] for g in got]), b.reshape(-1))
        assert np.array_equal(np.concatenate([g[1] for g in got]), q.reshape(-1))
        assert got[0][3][3] == b"r3"
    pairs = list(interleave(d + "/a.fastq", d + "/a.fastq.gz", batch_pairs=100))[0]
    assert len(pairs[2]) - 1 == 50 and np.array_equal(pairs[0][:60], pairs[0][60:120])


@pytest.mark.gpu
def test_cli_type_then_merge_end_to_end():
    from metamlst_amd.cli import main
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][6], size=200_000)
    b, q = synth.sample_reads(g, 20000)
    d = tempfile.mkdtemp()
    write_fastq(d + "/iso7.fastq.gz", b, q, gz=True)
    assert main(["type", d + "/iso7.fastq.gz", "-d", db.path, "-o", d + "/out", "--quiet", "--log"]) == 0
    assert os.path.exists(d + "/out/iso7.nfo")
    assert main(["merge", d + "/out", "-d", db.path]) == 0
    rep = open(d + "/out/merged/ecoli_report.txt").read().splitlines()
    assert rep[0].startswith("ST\tConfidence") and rep[1].split("\t") == ["7", "100.0", "iso7"]


@pytest.mark.gpu
def test_gpu_fastq_parser_equals_host_parser():
    """mlst_submit_fastq (FASTQ text parsed on the GPU) gives the statistics of the host-parsed path: LF and CRLF,
    with and without a final newline, ragged lengths, N bases, several chunks."""
    from metamlst_amd.engine import Engine, MlstError
    from metamlst_amd.fastq import text_chunks
    import fixtures as fx2
    db, idx = fx2.ecoli_small(40)
    rng = np.random.default_rng(3)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=60_000)
    recs = []
    for k in range(5000):
        L = int(rng.choice([150, 150, 101, 75, 36, 250, 320, 1]))
        at = int(rng.integers(0, len(g) - L))
        r = bytearray(g[at:at + L].tobytes())
        if k % 7 == 0 and L > 5:
            r[int(rng.integers(L))] = ord("N")
        q = bytes((rng.integers(2, 42, size=L)).astype(np.uint8) + 33)
        recs.append((b"r%d some comment" % k, bytes(r), q))
    eng = Engine(0)
    eng.load_reference(idx)
    eng.submit_reads(*synth.ragged_reads([r for _, r, _ in recs], [q for _, _, q in recs]))
    want = eng.stats()
    want_items = fx2.sorted_items(eng.items(1 << 16))
    d = tempfile.mkdtemp()
    for eol, final in ((b"\n", True), (b"\r\n", True), (b"\n", False)):
        text = eol.join(b"@" + n + eol + r + eol + b"+" + eol + q for n, r, q in recs) + (eol if final else b"")
        eng.reset_sample()
        assert eng.submit_fastq(text) == len(recs)
        fx2.assert_stats_equal(eng.stats(), want)
        assert np.array_equal(fx2.sorted_items(eng.items(1 << 16)), want_items)
    # chunked through the file helper (chunks cut after whole records), read indices continue across chunks
    path = d + "/x.fastq"
    open(path, "wb").write(b"\n".join(b"@" + n + b"\n" + r + b"\n+\n" + q for n, r, q in recs) + b"\n")
    eng.reset_sample()
    n = sum(eng.submit_fastq(c) for c in text_chunks(path, chunk_bytes=200_000))
    assert n == len(recs)
    fx2.assert_stats_equal(eng.stats(), want)
    # malformed input is refused
    eng.reset_sample()
    with pytest.raises(MlstError, match="4-line records"):
        eng.submit_fastq(b"@a\nACGT\n+\n")
    with pytest.raises(MlstError, match="malformed FASTQ"):
        eng.submit_fastq(b"@a\nACGT\n+\nIII\n")
    with pytest.raises(MlstError, match="longer than 320"):
        eng.submit_fastq(b"@a\n" + b"A" * 400 + b"\n+\n" + b"I" * 400 + b"\n")
    assert eng.submit_fastq(b"") == 0
