"""The pipelined typing loop as a product component (metamlst_amd/pipeline.py).

The reference types one sample per `metamlst.py` run (metamlst.py:96-289) into a folder that `metamlst-merge.py:93-107`
reads.  Eight samples through the pipeline -- four engines taking turns, each on its own share of the CUs, allele choice /
pile-up / consensus on the device -- must write byte for byte what eight serial runs of the host-driven path write
(statistics -> metamlst.py:133-151, 244 on the host -> pile-up -> .nfo line, --log table)."""
import os
import tempfile

import numpy as np
import pytest

from metamlst_amd import db as mdb
from metamlst_amd import synth
from metamlst_amd.engine import Engine
from metamlst_amd.index import load_index
from metamlst_amd.pipeline import TypingPipeline, make_engines
from metamlst_amd.typing import TypingArgs, log_table, type_sample

pytestmark = pytest.mark.gpu


def _samples(d, n_samples=8):
    db = synth.make_full_db(os.path.join(d, "m.db"), n_species=4, alleles_per_locus=40, n_profiles=12)
    out = []
    for k in range(n_samples):
        parts = []
        for j, sp in enumerate(db.species[k % 2:k % 2 + 2]):          # two species per sample, varying
            g, _ = synth.make_genome(db, sp, db.profiles[sp][(k + j) % 12], size=120_000, seed=70 + 5 * k + j)
            parts.append(synth.sample_reads(g, 5_000 + 1_500 * k, seed=90 + 5 * k + j))
        b = np.concatenate([p[0] for p in parts]); q = np.concatenate([p[1] for p in parts])
        perm = np.random.default_rng(k).permutation(len(b))
        out.append(synth.flatten_reads(b[perm], q[perm]))
    return db, out


def test_eight_samples_through_the_pipeline_equal_eight_serial_runs():
    with tempfile.TemporaryDirectory() as d:
        db, samples = _samples(d)
        idx = load_index(db.path)
        database = mdb.metaMLST_db(db.path)
        targs = TypingArgs()
        # serial, host-driven: one engine, three host round trips per sample
        eng = Engine(0)
        eng.load_reference(idx)
        want = []
        for fb, fq, off in samples:
            eng.reset_sample()
            eng.submit_reads(fb, fq, off)
            st = eng.stats()
            res = type_sample(idx, st, eng.pileup, database, "s", targs, out_dir=None)
            want.append(("".join(r.nfo_line for r in res if r.written), log_table(idx, st, targs, "s.fastq")))
        eng.close()
        assert all(w[0].count("\r\n") == 2 for w in want)                # both species of every sample written
        # pipelined: four engines on four CU shares, the device tail
        engines = make_engines(idx, 0, 4)
        pipe = TypingPipeline(engines, penalty=targs.penalty, stagger_s=0.5e-3)
        pipe.place(TypingPipeline.default_partitions(4))
        assert pipe.partitions == 4

        def feed(e, job):
            fb, fq, off = samples[job]
            e.submit_reads(fb, fq, off)

        def tail(job, st, chosen, letters):
            res = type_sample(idx, st, None, database, "s", targs, out_dir=None, typed=(chosen, letters))
            return job, "".join(r.nfo_line for r in res if r.written), log_table(idx, st, targs, "s.fastq")

        for round_ in range(2):                                          # the second round replays the engines' hipGraphs
            got = pipe.run(range(len(samples)), feed, tail, per_allele=True)
            assert [g[0] for g in got] == list(range(len(samples)))      # results come in submission order
            for (job, nfo, log), (w_nfo, w_log) in zip(got, want):
                assert nfo == w_nfo, (round_, job)
                assert log == w_log, (round_, job)
        # the fast host tail (no per-allele listing) on the same device results: the same lines
        def tail_fast(job, st, chosen, letters):
            res = type_sample(idx, st, None, database, "s", targs, out_dir=None, fast=True, typed=(chosen, letters))
            return "".join(r.nfo_line for r in res if r.written)
        assert pipe.run(range(len(samples)), feed, tail_fast, per_allele=False) == [w[0] for w in want]
        # fewer samples than engines, and none
        assert len(pipe.run(range(2), feed, tail)) == 2
        assert pipe.run([], feed, tail) == []
        # every engine fed by a thread of its own (what the folder mode does: feeds that block overlap): the same lines, and a
        # feeder's exception surfaces in the loop
        pipe.feed_threads = True
        got = pipe.run(range(len(samples)), feed, tail, per_allele=True)
        assert [(g[1], g[2]) for g in got] == want

        def bad_feed(e, job):
            if job == 3:
                raise ValueError("sample 3 cannot be read")
            feed(e, job)
        with pytest.raises(ValueError, match="sample 3"):
            pipe.run(range(len(samples)), bad_feed, tail)
        pipe.close()
        database.closeConnection()
