"""Shared seeded fixtures for the parity tests (small enough for the oracle to finish in seconds)."""
from __future__ import annotations

import functools
import os
import tempfile

import numpy as np

from metamlst_amd import synth
from metamlst_amd.index import load_index

_TMP = tempfile.mkdtemp(prefix="mlst_fix_")


@functools.lru_cache(maxsize=None)
def ecoli_small(alleles: int = 80, indel_every: int = 0, seed: int = synth.SEED):
    path = os.path.join(_TMP, "ecoli_%d_%d_%d.db" % (alleles, indel_every, seed))
    if os.path.exists(path):
        os.remove(path)
    db = synth.make_ecoli_db(path, alleles_per_locus=alleles, n_profiles=30, seed=seed, indel_every=indel_every)
    return db, load_index(path)


@functools.lru_cache(maxsize=None)
def multi_species(n_species: int = 3, alleles: int = 25):
    path = os.path.join(_TMP, "multi_%d_%d.db" % (n_species, alleles))
    if os.path.exists(path):
        os.remove(path)
    db = synth.make_full_db(path, n_species=n_species, alleles_per_locus=alleles, n_profiles=10)
    return db, load_index(path)


def isolate_reads(db, species, st_row, n_reads=20000, genome=200_000, seed=synth.SEED, read_len=150, err=0.001, mutate=None):
    g, starts = synth.make_genome(db, species, db.profiles[species][st_row], size=genome, seed=seed, mutate=mutate)
    b, q = synth.sample_reads(g, n_reads, read_len=read_len, seed=seed, err_rate=err)
    return synth.flatten_reads(b, q) + (g, starts)


def assert_stats_equal(a, b, counters=(0, 1, 4, 5, 6)):
    assert np.array_equal(a.sum_score, b.sum_score), "sum_score differs at %s" % np.nonzero(a.sum_score != b.sum_score)[0][:10]
    assert np.array_equal(a.n_hits, b.n_hits), "n_hits differs"
    assert np.array_equal(a.locus_len_sum, b.locus_len_sum), ("locus_len_sum", a.locus_len_sum, b.locus_len_sum)
    assert np.array_equal(a.locus_first, b.locus_first), ("locus_first", a.locus_first, b.locus_first)
    for c in counters:
        assert int(a.counters[c]) == int(b.counters[c]), ("counter", c, a.counters, b.counters)


def sorted_items(items):
    it = np.asarray(items, dtype=np.int64).reshape(-1, 5)
    return it[np.lexsort((it[:, 3], it[:, 2], it[:, 1], it[:, 0]))]
