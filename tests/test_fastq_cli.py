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
