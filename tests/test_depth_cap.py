"""Policy MLST_DEPTH_CAP as a switch (pysam's pileup(max_depth=8000) at metaMLST_functions.py:255-259).

pysam's cap depends on the order of the BAM file; the rule stated here does not: a column of a chosen allele sees the first `cap`
records that span it, records ordered by (read index, strand).  `orc_pileup_capped` (oracle/mlst_oracle.c) states it read by
read; the engine finds, per column, the key of the cap-th record by a bitwise search over counting passes and piles up what each
column sees (`mlst_set_depth_cap`).  Default off: every other test of this repository runs without a cap."""
import os
import tempfile

import numpy as np
import pytest

import fixtures as fx
import oracle_lib
from metamlst_amd import synth
from metamlst_amd.engine import default_params
from metamlst_amd.index import load_index
from metamlst_amd.typing import consensus_from_counts, pick_alleles_fast


def _chosen(idx, orc):
    return sorted(pick_alleles_fast(idx, orc.stats(), 100).values())


def test_oracle_cap_is_the_first_records_of_every_column():
    """Identical reads over one stretch of one locus: with cap = 3 every column they span counts exactly three, with a cap above the
    depth the capped pile-up is the plain one, and the depth reported is the number of records whatever the cap."""
    db, idx = fx.ecoli_small(20)
    g, starts = synth.make_genome(db, "ecoli", db.profiles["ecoli"][1], size=40_000)
    gene0 = db.loci["ecoli"][0][0]
    at = int(starts[gene0]) + 40
    read = g[at:at + 150].tobytes()
    other = g[at + 60:at + 210].tobytes()
    reads = [read] * 7 + [other] * 2
    fb, fq, off = synth.ragged_reads(reads, [b"I" * 150] * len(reads))
    orc = oracle_lib.Oracle(idx)
    orc.submit_reads(fb, fq, off)
    chosen = _chosen(idx, orc)
    plain = orc.pileup(chosen)
    a0 = [a for a in chosen if plain[a].sum() > 0]
    assert len(a0) == 1
    a0 = a0[0]
    depth = {}
    capped = orc.pileup(chosen, depth_cap=3, depth_out=depth)
    cov, cov_c = plain[a0].sum(axis=1), capped[a0].sum(axis=1)
    assert cov.max() == 9 and set(np.unique(cov)) == {0, 2, 7, 9}
    assert np.array_equal(depth[a0], cov)                              # Phred 40 everywhere: every spanning record counts a base
    assert np.array_equal(cov_c, np.minimum(cov, 3))
    # the first three records of every column are copies of the first read (lower read indices): one base, three times -- also
    # where the other two reads agree with it or not
    deep = np.nonzero(cov >= 7)[0]
    assert (capped[a0][deep].max(axis=1) == 3).all()
    assert np.array_equal(capped[a0][deep].argmax(axis=1), plain[a0][deep].argmax(axis=1))
    big = orc.pileup(chosen, depth_cap=1000)
    for a in chosen:
        assert np.array_equal(big[a], plain[a])


def test_oracle_cap_monotone_on_a_deep_sample():
    db, idx = fx.ecoli_small(40, indel_every=5)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 2, n_reads=6000, genome=60_000)
    orc = oracle_lib.Oracle(idx)
    orc.submit_reads(fb, fq, off)
    chosen = _chosen(idx, orc)
    plain = orc.pileup(chosen)
    prev = None
    for cap in (2, 5, 11, 10_000):
        depth = {}
        c = orc.pileup(chosen, depth_cap=cap, depth_out=depth)
        for a in chosen:
            assert (c[a] <= plain[a]).all() and (c[a].sum(axis=1) <= cap).all()
            assert (c[a].sum(axis=1) <= np.minimum(depth[a], cap)).all()
            if prev is not None:
                assert (prev[a] <= c[a]).all()                         # a larger cap sees a superset of the records
        prev = c
    for a in chosen:
        assert np.array_equal(prev[a], plain[a])


def _engine(idx, params=None):
    from metamlst_amd.engine import Engine
    eng = Engine(0, params)
    eng.load_reference(idx)
    return eng


def _compare(eng, orc, idx, caps):
    chosen = _chosen(idx, orc)
    plain = orc.pileup(chosen)
    for cap in caps:
        eng.set_depth_cap(cap)
        want = orc.pileup(chosen, depth_cap=cap)
        got = eng.pileup(chosen)
        assert set(got) == set(want)
        for a in chosen:
            assert np.array_equal(got[a], want[a]), ("cap %d: pile-up differs for allele %d" % (cap, a), np.nonzero((got[a] != want[a]).any(axis=1))[0][:8])
        cons = eng.consensus(chosen)
        for a in chosen:
            assert cons[a].decode() == "".join(consensus_from_counts(want[a]))
    assert any((orc.pileup(chosen, depth_cap=caps[0])[a] != plain[a]).any() for a in chosen), "the cap never bit"
    eng.set_depth_cap(0)
    got = eng.pileup(chosen)
    for a in chosen:
        assert np.array_equal(got[a], plain[a])
    return chosen


@pytest.mark.gpu
def test_engine_capped_pileup_equals_the_oracle_single_end_indels_and_two_submissions():
    db, idx = fx.ecoli_small(60, indel_every=4)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 2, n_reads=15_000, genome=150_000)
    eng, orc = _engine(idx), oracle_lib.Oracle(idx)
    half = 7_001                                                        # two submissions: the read index runs on (read_base)
    eng.submit_reads(fb[:off[half]], fq[:off[half]], off[:half + 1])
    eng.submit_reads(fb[off[half]:], fq[off[half]:], off[half:] - off[half])
    orc.submit_reads(fb, fq, off)
    s = eng.stats()
    fx.assert_stats_equal(s, orc.stats())
    assert s.counters[6] > 0, "no pair reached the banded Smith-Waterman"
    _compare(eng, orc, idx, (4, 9, 1, 100_000))
    eng.close()


@pytest.mark.gpu
def test_engine_capped_pileup_always_banded_and_paired():
    p = default_params()
    p.gap_trigger_mm = -1                                               # every record through the banded SW and its traceback
    db, idx = fx.ecoli_small(12, indel_every=3)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 1, n_reads=3000, genome=60_000)
    eng, orc = _engine(idx, p), oracle_lib.Oracle(idx, p)
    eng.submit_reads(fb, fq, off)
    orc.submit_reads(fb, fq, off)
    _compare(eng, orc, idx, (3, 6))
    eng.close()
    with tempfile.TemporaryDirectory() as d:                            # mates: two read indices, both may span a column
        db = synth.make_ecoli_db(os.path.join(d, "p.db"), alleles_per_locus=30, n_profiles=20)
        g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][4], size=80_000, seed=11)
        b, q = synth.sample_pairs(g, n_pairs=4000, seed=12)
        fb, fq, off = synth.flatten_reads(b, q)
        idx = load_index(db.path)
        eng, orc = _engine(idx), oracle_lib.Oracle(idx)
        eng.submit_reads(fb, fq, off, paired=True)
        orc.submit_reads(fb, fq, off, paired=True)
        _compare(eng, orc, idx, (5, 12))
        eng.close()


@pytest.mark.gpu
def test_capped_typing_tail_on_the_device():
    """typing_enqueue under a cap: the letters are the majority rule over the capped oracle counts of the alleles the device chose;
    without the cap again, those of the plain counts (the replayed graph of the uncapped tail is rebuilt)."""
    db, idx = fx.ecoli_small(50)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 3, n_reads=12_000, genome=120_000)
    eng, orc = _engine(idx), oracle_lib.Oracle(idx)
    orc.submit_reads(fb, fq, off)
    for cap in (0, 0, 0, 5, 5, 0, 0):                                   # (the third uncapped round replays the graph; then in and out of the cap)
        eng.reset_sample()
        eng.set_depth_cap(cap)
        eng.submit_reads(fb, fq, off)
        eng.typing_enqueue(penalty=100)
        st, chosen, letters = eng.typing_fetch()
        ch = sorted(chosen.values())
        assert len(ch) == 7
        want = orc.pileup(ch, depth_cap=cap)
        for a in ch:
            assert letters[a].decode() == "".join(consensus_from_counts(want[a])), (cap, a)
    eng.close()


@pytest.mark.gpu
def test_cli_depth_cap_writes_the_nfo_line_of_the_capped_oracle_pileup(tmp_path):
    """`cli type SAMPLE --depth-cap N` (one sample, and a folder of two through the pipelined loop): the .nfo line is the one
    the host statement of metamlst.py:133-289 writes over the ORACLE's capped pile-up; without the switch, over the plain one;
    and the two differ on this sample (the cap drops mismatching late reads from deep columns)."""
    import subprocess
    import sys
    from metamlst_amd import db as mdb
    from metamlst_amd.typing import TypingArgs, type_sample
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    db, idx = fx.ecoli_small(40)
    # a novel allele: SNPs planted in the genome at columns the reads cover deeply; half of the reads of the sample come first
    # WITHOUT them (another isolate's reads), so what the first N records of a column say differs from what all of them say
    g0, starts = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=90_000)
    gene = db.loci["ecoli"][0][0]
    seq = synth.allele_sequence(db.path, "ecoli", gene, int(db.profiles["ecoli"][2][0]))
    muts = [(p, "A" if seq[p] != "A" else "C") for p in (60, 200, 333)]
    g1, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=90_000, mutate={gene: muts})
    b0, q0 = synth.sample_reads(g0, 1500, seed=5, err_rate=0.0)      # 2.5x: the first records of every column
    b1, q1 = synth.sample_reads(g1, 9000, seed=6, err_rate=0.0)      # 15x: the majority
    bases, quals = np.concatenate([b0, b1]), np.concatenate([q0, q1])
    fq = tmp_path / "s.fastq"
    with open(fq, "wb") as f:
        for k in range(len(bases)):
            f.write(b"@r%d\n" % k + bases[k].tobytes() + b"\n+\n" + quals[k].tobytes() + b"\n")
    fb, fq_q, off = synth.flatten_reads(bases, quals)
    orc = oracle_lib.Oracle(idx)
    orc.submit_reads(fb, fq_q, off)
    database = mdb.metaMLST_db(db.path)
    want = {}
    for cap in (0, 2):
        res = type_sample(idx, orc.stats(), lambda ch, cap=cap: orc.pileup(ch, depth_cap=cap), database, "s", TypingArgs(), out_dir=None)
        want[cap] = "".join(r.nfo_line for r in res if r.written)
        assert want[cap].count("\r\n") == 1
    assert want[0] != want[2]
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")

    def cli(args):
        r = subprocess.run([sys.executable, "-m", "metamlst_amd.cli"] + args, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

    for cap in (0, 2):
        out = tmp_path / ("out%d" % cap)
        cli(["type", str(fq), "-d", db.path, "-o", str(out), "--quiet"] + (["--depth-cap", str(cap)] if cap else []))
        assert open(out / "s.nfo", newline="").read() == want[cap], cap
    folder = tmp_path / "reads"
    folder.mkdir()
    for name in ("a", "b"):
        (folder / (name + ".fastq")).write_bytes(fq.read_bytes())
    out = tmp_path / "outf"
    cli(["type", str(folder), "-d", db.path, "-o", str(out), "--quiet", "--depth-cap", "2"])
    for name in ("a", "b"):
        assert open(out / (name + ".nfo"), newline="").read() == want[2].replace("\ts\t", "\t%s\t" % name)
    database.closeConnection()
