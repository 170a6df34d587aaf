"""The persistent index cache next to the database (the reference skips its index build when <idx>.1.bt2
exists: /root/reference/metamlst-index.py:224-225) and the lazy look-up tables of the host tail: same answers as without."""
import os
import sqlite3
import time

import numpy as np

from metamlst_amd import db as mdb
from metamlst_amd import synth
from metamlst_amd.index import _cache_path, load_index


def _same(a, b):
    assert a.species == b.species and a.loci == b.loci
    for f in ("locus_id", "species_id", "allele_no", "rec_id", "off", "ascii_concat", "locus_begin", "locus_count", "locus_species", "locus_maxlen"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_index_cache_round_trip_and_invalidation(tmp_path):
    sdb = synth.make_full_db(str(tmp_path / "f.db"), n_species=5, alleles_per_locus=12, n_profiles=6)
    fresh = load_index(sdb.path, cache=False)
    assert not os.path.exists(_cache_path(sdb.path))
    first = load_index(sdb.path)                      # builds and stores
    assert os.path.exists(_cache_path(sdb.path))
    again = load_index(sdb.path)                      # from the file
    _same(fresh, first); _same(fresh, again)
    # a species filter never reads or writes the cache
    flt = load_index(sdb.path, [sdb.species[1]])
    assert flt.species == [sdb.species[1]]
    # the database changes -> the cache no longer names it
    time.sleep(0.01)
    conn = sqlite3.connect(sdb.path)
    sp, gene = fresh.loci[0]
    conn.execute("INSERT INTO alleles (bacterium,gene,sequence,alignedSequence,alleleVariant) VALUES (?,?,?,?,?)", (sp, gene, "ACGT" * 100, "", 9999))
    conn.commit(); conn.close()
    changed = load_index(sdb.path)
    assert changed.n_alleles == fresh.n_alleles + 1
    _same(changed, load_index(sdb.path, cache=False))
    os.environ["MLST_INDEX_CACHE"] = "0"
    try:
        os.unlink(_cache_path(sdb.path))
        load_index(sdb.path)
        assert not os.path.exists(_cache_path(sdb.path))
    finally:
        os.environ.pop("MLST_INDEX_CACHE")


def test_lazy_dbcache_equals_the_sql_helpers(tmp_path):
    sdb = synth.make_full_db(str(tmp_path / "g.db"), n_species=4, alleles_per_locus=9, n_profiles=5)
    # a duplicated sequence under a later row and a row without a sequence: first row wins / not in the index
    conn = sqlite3.connect(sdb.path)
    sp = sdb.species[2]
    gene = sdb.loci[sp][3][0]
    seq3 = synth.allele_sequence(sdb.path, sp, gene, 3)
    conn.execute("INSERT INTO alleles (bacterium,gene,sequence,alignedSequence,alleleVariant) VALUES (?,?,?,?,?)", (sp, gene, seq3, "", 77))
    conn.execute("INSERT INTO alleles (bacterium,gene,sequence,alignedSequence,alleleVariant) VALUES (?,?,?,?,?)", (sp, gene, "", "", 78))
    conn.commit(); conn.close()
    idx = load_index(sdb.path, cache=False)
    database = mdb.metaMLST_db(sdb.path)
    with_index, without = mdb.DbCache(database.conn, idx), mdb.DbCache(database.conn)
    probes = [(sp, seq3), (sp, seq3[:-1] + ("A" if seq3[-1] != "A" else "C")), (sp, ""), ("nobody", seq3), (sdb.species[0], seq3)]
    for s2 in sdb.species:
        g2 = sdb.loci[s2][0][0]
        probes.append((s2, synth.allele_sequence(sdb.path, s2, g2, 2)))
    for b, q in probes:
        want = mdb.sequenceExists(database.conn, b, q)
        for c in (with_index, without):
            assert c.sequenceExists(b, q) == want
            assert c.sequenceFind(b, q) == mdb.sequenceFind(database.conn, b, q)
            if want:
                assert c.sequenceLocate(b, q) == mdb.sequenceLocate(database.conn, b, q)
    labels = ["%s_%s_%d" % (sp, g, 1 + k % 5) for k, (g, _) in enumerate(sdb.loci[sp])]
    for c in (with_index, without):
        assert sorted(c.defineProfile(labels)) == sorted(mdb.defineProfile(database.conn, labels))      # (profiles tied on the count: SQLite orders them arbitrarily, Q9)
    assert set(with_index._seq_sp) <= set(sdb.species) | {"nobody"}


import pytest


@pytest.mark.gpu
def test_reference_cache_file_gives_the_same_engine(tmp_path):
    """mlst_set_reference_cache: the host index is built once, stored, and a later mlst_load_reference (the in-process cache
    dropped, as in a new process) reads it back: same statistics, work items and pile-up on the same reads; a file that
    belongs to other inputs is ignored and replaced."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fixtures as fx
    from metamlst_amd.engine import Engine, load_library
    from metamlst_amd.typing import pick_alleles_fast
    lib = load_library()
    sdb = synth.make_ecoli_db(str(tmp_path / "e.db"), alleles_per_locus=40, n_profiles=10)
    idx = load_index(sdb.path, cache=False)
    g, _ = synth.make_genome(sdb, "ecoli", sdb.profiles["ecoli"][2], size=120_000)
    b, q = synth.sample_reads(g, 15_000)
    fb, fq, off = synth.flatten_reads(b, q)
    ref = str(tmp_path / "e.db.mlstref")

    def run(cache_path):
        lib.mlst_release_index_cache()
        e = Engine(0)
        e.load_reference(idx, cache_path=cache_path)
        e.submit_reads(fb, fq, off)
        s = e.stats()
        ch = sorted(pick_alleles_fast(idx, s, 100).values())
        pl = e.pileup(ch)
        items = fx.sorted_items(e.items(1 << 16))
        e.close()
        return s, ch, pl, items

    try:
        s0, ch0, pl0, it0 = run("")                 # built, nothing stored
        assert not os.path.exists(ref)
        s1, ch1, pl1, it1 = run(ref)                # built and stored
        assert os.path.getsize(ref) > 10_000
        t_file = os.path.getmtime(ref)
        s2, ch2, pl2, it2 = run(ref)                # read back
        assert os.path.getmtime(ref) == t_file
        for s, ch, pl, it in ((s1, ch1, pl1, it1), (s2, ch2, pl2, it2)):
            fx.assert_stats_equal(s, s0)
            assert ch == ch0 and np.array_equal(it, it0)
            for a in ch0:
                assert np.array_equal(pl[a], pl0[a])
        # another database under the same file name: the key in the header does not fit, the file is replaced
        sdb2 = synth.make_ecoli_db(str(tmp_path / "e2.db"), alleles_per_locus=25, n_profiles=5, seed=99)
        idx2 = load_index(sdb2.path, cache=False)
        lib.mlst_release_index_cache()
        e = Engine(0)
        e.load_reference(idx2, cache_path=ref)
        assert e.index_bytes()[0] > 0
        e.close()
        lib.mlst_release_index_cache()
        e = Engine(0)
        e.load_reference(idx, cache_path=ref)       # (and back: rebuilt, not misread)
        e.submit_reads(fb, fq, off)
        fx.assert_stats_equal(e.stats(), s0)
        e.close()
    finally:
        lib.mlst_set_reference_cache(None)
