"""Literal parity with the reference's aligner, wherever it exists: bowtie2 + samtools
are in neither the reference tree nor this image, so on most machines this test SKIPS.  Where they are installed it runs
the documented command (README.md:20) on a cfg1-style isolate, feeds the BAM to the --alignments path and requires the
ST-relevant outcome of the FASTQ path to agree; the record-level deviation is printed."""
import json
import os
import tempfile

import pytest

import fixtures as fx
from metamlst_amd import db as mdb
from metamlst_amd import literal, synth


def test_without_the_tools_the_leg_reports_that_it_was_skipped(monkeypatch):
    monkeypatch.setattr(literal.shutil, "which", lambda name: None)
    assert literal.tools() is None
    assert "skipped" in literal.literal_parity(None, None, None, "/nonexistent.db", "/nonexistent.fastq")


@pytest.mark.gpu
@pytest.mark.skipif(literal.tools() is None, reason="bowtie2 / bowtie2-build / samtools not on PATH")
def test_fastq_path_agrees_with_bowtie2_on_an_isolate():
    from metamlst_amd.engine import Engine
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][5], size=200_000)
    b, q = synth.sample_reads(g, 30_000)
    d = tempfile.mkdtemp(prefix="mlst_lit_")
    fq = os.path.join(d, "iso.fastq")
    with open(fq, "wb") as f:
        for k in range(len(b)):
            f.write(b"@r%d\n" % k + b[k].tobytes() + b"\n+\n" + q[k].tobytes() + b"\n")
    eng = Engine(0)
    eng.load_reference(idx)
    out = literal.literal_parity(eng, idx, mdb.metaMLST_db(db.path), db.path, fq, threads=4, keep_dir=d)
    print(json.dumps(out))
    assert "error" not in out, out
    assert out["chosen_alleles_equal"] and out["nfo_lines"]["gpu"] == out["nfo_lines"]["bowtie2"] == 1
