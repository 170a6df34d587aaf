"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle, bit exact."""
import numpy as np
import pytest

import fixtures as fx
from metamlst_amd.engine import MlstError
import oracle_lib
from metamlst_amd import synth
from metamlst_amd.engine import Engine, default_params
from metamlst_amd.typing import build_consensus, pick_alleles_fast

pytestmark = pytest.mark.gpu


def both(idx, params=None):
    eng = Engine(0, params)
    eng.load_reference(idx)
    return eng, oracle_lib.Oracle(idx, params)


def run_both(eng, orc, fb, fq, off):
    eng.reset_sample()
    eng.submit_reads(fb, fq, off)
    orc.submit_reads(fb, fq, off)
    s = eng.stats()
    so, items_o = orc.stats(want_items=1 << 18)
    fx.assert_stats_equal(s, so)
    assert np.array_equal(fx.sorted_items(eng.items(1 << 18)), fx.sorted_items(items_o))
    return s, so


def check_pileup(eng, orc, idx, s):
    from metamlst_amd.typing import consensus_from_counts
    chosen = sorted(pick_alleles_fast(idx, s, 100).values())
    pc, po = eng.pileup(chosen), orc.pileup(chosen)
    assert set(pc) == set(po)
    for a in pc:
        assert np.array_equal(pc[a], po[a]), "pileup differs for allele %d" % a
    cons = eng.consensus(chosen)            # GPU majority rule == host majority rule on the oracle's counts
    for a in chosen:
        assert cons[a].decode() == "".join(consensus_from_counts(po[a])), "consensus differs for allele %d" % a
    return chosen, pc


def test_extension_kernel_choice_and_addition_forms_agree_with_the_oracle():
    """k_extend scores block haplotypes (loci up to MLST_EXT_HAP_MAX alleles), k_extend_pairs aligns pair by pair; the fast pass adds
    count and sum in one 64-bit addition while a submission holds fewer than 2^24 items and in two otherwise.  Every combination
    = the oracle (align_pair of oracle/mlst_oracle.c is the specification), on a database with indel alleles (banded SW, the
    aligned-span test) and on an isolate."""
    import os
    for alleles, indel in ((80, 0), (60, 7)):
        db, idx = fx.ecoli_small(alleles, indel_every=indel)
        fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 3, n_reads=12_000)
        orc = oracle_lib.Oracle(idx)
        orc.submit_reads(fb, fq, off)
        so, items_o = orc.stats(want_items=1 << 18)
        for env, max_items in (({}, 0), ({"MLST_EXT_HAP_MAX": "0"}, 0), ({"MLST_EXT_LDS_KB": "0"}, 0), ({"MLST_EXT_HAP_MAX": "40"}, 0), ({}, 1 << 24),
                               ({"MLST_EXT_THREADS": "256"}, 0)):      # (the last: four waves per work item in the haplotype kernel)
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                p = default_params()
                p.max_items = max_items
                eng = Engine(0, p)
                eng.load_reference(idx)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            info = eng.extend_info()
            assert info["loci"] == (idx.n_loci if (not env or "MLST_EXT_THREADS" in env) else 0), (env, info)      # (40 < every locus' allele count)
            for _ in range(2):                                                       # the second submission replays the hipGraph
                eng.reset_sample()
                eng.submit_reads(fb, fq, off)
                fx.assert_stats_equal(eng.stats(), so)
            assert np.array_equal(fx.sorted_items(eng.items(1 << 18)), fx.sorted_items(items_o))
            eng.close()


def test_ecoli_isolate_pass1_and_pileup():
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 5)
    eng, orc = both(idx)
    s, _ = run_both(eng, orc, fb, fq, off)
    assert s.counters[0] > 1000
    chosen, pc = check_pileup(eng, orc, idx, s)
    # the closest allele may be a rounded-average tie (Q5); the consensus must spell the planted allele
    for a in chosen:
        gene = idx.loci[int(idx.locus_id[a])][1]
        true_no = int(db.profiles["ecoli"][5][[g for g, _ in db.loci["ecoli"]].index(gene)])
        rec = build_consensus({idx.label(a): idx.sequence(a)}, {idx.label(a): pc[a]})[0]
        assert rec.seq.upper() == synth.allele_sequence(db.path, "ecoli", gene, true_no)


def test_indel_alleles_exercise_banded_sw():
    db, idx = fx.ecoli_small(60, indel_every=4)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 2, n_reads=15000, genome=150_000)
    eng, orc = both(idx)
    s, _ = run_both(eng, orc, fb, fq, off)
    assert s.counters[6] > 0, "no pair reached the banded Smith-Waterman"
    check_pileup(eng, orc, idx, s)


def test_always_banded_policy():
    p = default_params()
    p.gap_trigger_mm = -1
    db, idx = fx.ecoli_small(12, indel_every=3)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 1, n_reads=3000, genome=60_000)
    eng, orc = both(idx, p)
    s, _ = run_both(eng, orc, fb, fq, off)
    check_pileup(eng, orc, idx, s)


def test_ragged_reads_with_n_and_short_reads():
    db, idx = fx.ecoli_small(40)
    rng = np.random.default_rng(7)
    g, starts = synth.make_genome(db, "ecoli", db.profiles["ecoli"][0], size=60_000)
    reads, quals = [], []
    for k in range(6000):
        L = int(rng.integers(1, 301))
        at = int(rng.integers(0, len(g) - L))
        r = bytearray(g[at:at + L].tobytes())
        if k % 2:
            r = bytearray(synth._COMP[np.frombuffer(bytes(r), np.uint8)[::-1]].tobytes())
        q = bytearray(rng.integers(2, 42, size=L).astype(np.uint8) + 33)
        if k % 5 == 0 and L > 3:
            for p in rng.integers(0, L, size=2):
                r[p] = ord("N")
                q[p] = 2 + 33
        reads.append(bytes(r))
        quals.append(bytes(q))
    fb, fq, off = synth.ragged_reads(reads, quals)
    eng, orc = both(idx)
    s, _ = run_both(eng, orc, fb, fq, off)
    assert s.counters[0] > 0
    check_pileup(eng, orc, idx, s)


def test_empty_batch_and_no_hits():
    db, idx = fx.ecoli_small(40)
    eng, orc = both(idx)
    eng.submit_reads(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    s = eng.stats()
    assert s.sum_score.sum() == 0 and s.counters[0] == 0
    rng = np.random.default_rng(3)
    b = synth._ACGT[rng.integers(0, 4, size=(2000, 150))]
    q = np.full_like(b, 40 + 33)
    fb, fq, off = synth.flatten_reads(b, q)
    run_both(eng, orc, fb, fq, off)


def test_batches_equal_single_submission():
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 7, n_reads=12000)
    eng, orc = both(idx)
    s1, _ = run_both(eng, orc, fb, fq, off)
    eng.reset_sample()
    n = len(off) - 1
    for lo, hi in ((0, 5000), (5000, 5001), (5001, n)):
        o = off[lo:hi + 1]
        eng.submit_reads(fb[int(o[0]):int(o[-1])], fq[int(o[0]):int(o[-1])], o - o[0])
    fx.assert_stats_equal(eng.stats(), s1)


def test_multi_species_metagenome():
    db, idx = fx.multi_species(3, 25)
    parts = []
    for k, sp in enumerate(db.species):
        g, _ = synth.make_genome(db, sp, db.profiles[sp][k], size=80_000, seed=100 + k)
        parts.append(synth.sample_reads(g, 6000, seed=200 + k))
    b = np.concatenate([p[0] for p in parts])
    q = np.concatenate([p[1] for p in parts])
    perm = np.random.default_rng(5).permutation(len(b))
    fb, fq, off = synth.flatten_reads(b[perm], q[perm])
    eng, orc = both(idx)
    s, _ = run_both(eng, orc, fb, fq, off)
    chosen, pc = check_pileup(eng, orc, idx, s)
    by_locus = {idx.loci[int(idx.locus_id[a])]: a for a in chosen}
    for k, sp in enumerate(db.species):
        for (gname, _), al in zip(db.loci[sp], db.profiles[sp][k]):
            a = by_locus[(sp, gname)]
            rec = build_consensus({idx.label(a): idx.sequence(a)}, {idx.label(a): pc[a]})[0]
            assert rec.seq.upper() == synth.allele_sequence(db.path, sp, gname, int(al))


def test_hamming_matches_stringdiff():
    db, idx = fx.ecoli_small(80, indel_every=5)
    eng, orc = both(idx)
    rng = np.random.default_rng(11)
    for locus in (0, 3, 6):
        base = idx.sequence(int(idx.locus_begin[locus]) + 17)
        for trial in range(4):
            s = bytearray(base.encode())
            for p in rng.integers(0, len(s), size=trial * 3):
                s[p] = ord("ACGT"[int(rng.integers(4))])
            if trial == 2:
                s = s[:-25]
            if trial == 3:
                s = s + b"ACGTACGT"
            q = bytes(s)
            assert np.array_equal(eng.hamming_all(locus, q), orc.hamming_all(locus, q))
            assert eng.hamming_le(locus, q, 5) == orc.hamming_le(locus, q, 5)
    assert eng.hamming_le(0, b"", 0)[1] == int(idx.locus_count[0])     # zip with '' -> 0 mismatches


def test_sharded_engines_reduce_to_the_single_engine_result():
    """Multi-GPU data path on one GPU: two engines each see half of the reads (as two ranks would), their
    statistics are summed / min-reduced through mlst_export_stats_device + mlst_import_stats_device and the
    pileups through mlst_pileup_device; the result must equal one engine that saw every read."""
    import torch
    from metamlst_amd.dist import DeviceStatsPort, shard_range, split_counts
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 4, n_reads=14001)
    dev = torch.device("cuda", 0)
    whole = Engine(0)
    whole.load_reference(idx)
    whole.submit_reads(fb, fq, off)
    s_all = whole.stats()
    n = len(off) - 1
    shards, ports = [], []
    for r in range(2):
        lo, hi = shard_range(n, r, 2)
        e = Engine(0)
        e.load_reference(idx)
        o = off[lo:hi + 1]
        e.set_read_index_base(lo)     # first-seen order (Q6) must be global across shards
        e.submit_reads(fb[int(o[0]):int(o[-1])], fq[int(o[0]):int(o[-1])], o - o[0])
        shards.append(e)
        ports.append(DeviceStatsPort(e, dev))
    n_sum, n_min = ports[0].flat_sizes()
    sums = [torch.empty(n_sum, dtype=torch.int64, device=dev) for _ in range(2)]
    mins = [torch.empty(max(1, n_min), dtype=torch.int64, device=dev) for _ in range(2)]
    for p, ts, tm in zip(ports, sums, mins):
        p.export_stats(ts, tm)
    tot, mn = sums[0] + sums[1], torch.minimum(mins[0], mins[1])
    for p in ports:
        p.import_stats(tot, mn)
    for e in shards:
        s = e.stats()
        assert np.array_equal(s.sum_score, s_all.sum_score) and np.array_equal(s.n_hits, s_all.n_hits)
        assert np.array_equal(s.locus_len_sum, s_all.locus_len_sum) and np.array_equal(s.locus_first, s_all.locus_first)
        assert all(int(s.counters[k]) == int(s_all.counters[k]) for k in (0, 1, 2, 6))
    chosen = sorted(pick_alleles_fast(idx, shards[0].stats(), 100).values())
    assert chosen == sorted(pick_alleles_fast(idx, s_all, 100).values())
    n_cols = sum(int(idx.off[a + 1] - idx.off[a]) for a in chosen)
    parts = []
    for p in ports:
        t = torch.zeros(n_cols * 4, dtype=torch.int32, device=dev)
        assert p.pileup_into(chosen, t) == n_cols
        parts.append(t)
    summed = parts[0] + parts[1]
    merged = split_counts(idx, chosen, summed.cpu().numpy().view(np.uint32).reshape(-1, 4))
    ref = whole.pileup(chosen)
    for a in chosen:
        assert np.array_equal(merged[a], ref[a])
    # majority rule on the reduced counts, on the device (what rank 0 does in the multi-GPU flow)
    letters = ports[0].consensus_from_counts(summed, n_cols)
    assert letters == b"".join(whole.consensus(chosen)[a] for a in chosen)


@pytest.mark.parametrize("kind,waves", [("global", "16"), ("routed", "16"), ("routed", "8")])
def test_big_database_sieves_on_a_small_database(monkeypatch, kind, waves):
    """Large databases skip the LDS first-level bitmap; force those paths on a small one (MLST_SIEVE): the single-kernel
    sieve with the global bitmap and the CU-routed sieve (k_route -> k_route_probe -> k_flag_compact: the one databases
    beyond the LDS bitmaps get by default; with 16- and 8-wave producer workgroups)."""
    monkeypatch.setenv("MLST_SIEVE", kind)
    monkeypatch.setenv("MLST_ROUTE_WAVES", waves)
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 8, n_reads=9000)
    for _once in (0,):
        eng, orc = both(idx)
        assert eng.sieve_info()["kind"] == kind
        s, _ = run_both(eng, orc, fb, fq, off)
        check_pileup(eng, orc, idx, s)
        # ragged lengths (fewer seeds than slots, reads shorter than a seed) and N bases through the same path
        rng = np.random.default_rng(5)
        reads, quals = [], []
        for k in range(0, 3000):
            o = int(off[k]); L = int(rng.integers(1, 151))
            r = bytearray(fb[o:o + L].tobytes())
            if k % 9 == 0:
                r[int(rng.integers(L))] = ord("N")
            reads.append(bytes(r)); quals.append(fq[o:o + L].tobytes())
        fbr, fqr, offr = synth.ragged_reads(reads, quals)
        orc2 = oracle_lib.Oracle(idx)
        eng.reset_sample()
        eng.submit_reads(fbr, fqr, offr); orc2.submit_reads(fbr, fqr, offr)
        fx.assert_stats_equal(eng.stats(), orc2.stats())


def test_alleles_with_ambiguity_codes_use_the_n_mask_paths():
    """Database alleles containing N / IUPAC codes: --np 1 penalty, counts in XM, never a seed."""
    import sqlite3
    import tempfile
    from metamlst_amd.index import load_index
    d = tempfile.mkdtemp()
    db = synth.make_ecoli_db(d + "/n.db", alleles_per_locus=30, n_profiles=10, seed=77, indel_every=6)
    conn = sqlite3.connect(db.path)
    rng = np.random.default_rng(2)
    rows = conn.execute("SELECT recID, sequence FROM alleles").fetchall()
    for rid, seq in rows:
        if rid % 3 == 0:
            s = list(seq)
            for p in rng.integers(0, len(s), size=int(rng.integers(1, 4))):
                s[p] = "NRYK"[int(rng.integers(4))]
            conn.execute("UPDATE alleles SET sequence=? WHERE recID=?", ("".join(s), rid))
    conn.commit()
    conn.close()
    idx = load_index(db.path)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 3, n_reads=12000, genome=120_000)
    for trig in (12, -1):
        p = default_params()
        p.gap_trigger_mm = trig
        eng, orc = both(idx, p)
        n = 12000 if trig > 0 else 1500
        o = off[:n + 1]
        s, _ = run_both(eng, orc, fb[:int(o[-1])], fq[:int(o[-1])], o)
        check_pileup(eng, orc, idx, s)


def test_paired_end_five_fold_coverage_cfg5():
    """cfg5 of SURVEY.md 8(d): 2 x 150 bp pairs, insert N(300, 30), loci at ~5x.  The documented pipeline aligns mates as
    unpaired reads (bowtie2 -U): every mate gets its own records.  Sharing a QNAME matters in one place, the dictionary
    sequenceBank[locus][QNAME] = len(SEQ) (metamlst.py:127, Q3): of two mates with accepted records on a locus only the
    second one's length stays.  Engine (paired=1: candidates in pairs, mate links, k_locus) = oracle, and the per-locus
    sums are smaller than those of the same reads submitted as single reads by exactly the first mates' lengths."""
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=100_000)
    b, q = synth.sample_pairs(g, n_pairs=int(100_000 * 5 / 300))
    fb, fq, off = synth.flatten_reads(b, q)
    eng, orc = both(idx)
    eng.reset_sample()
    eng.submit_reads(fb, fq, off, paired=True)
    orc.submit_reads(fb, fq, off, paired=True)
    s = eng.stats()
    so, items_o = orc.stats(want_items=1 << 16)
    fx.assert_stats_equal(s, so)
    chosen, pc = check_pileup(eng, orc, idx, s)
    holes = sum(int((pc[a].sum(axis=1) == 0).sum()) for a in chosen)
    assert holes > 0, "5x coverage should leave some columns for the gap-fill path"
    # the same reads as single reads: identical hits and scores, larger length sums (both mates of a pair counted)
    eng.reset_sample()
    eng.submit_reads(fb, fq, off, paired=False)
    orc.submit_reads(fb, fq, off, paired=False)
    s1, so1 = eng.stats(), orc.stats()
    fx.assert_stats_equal(s1, so1)
    assert np.array_equal(s1.sum_score, s.sum_score) and np.array_equal(s1.n_hits, s.n_hits)
    assert (s1.locus_len_sum >= s.locus_len_sum).all() and int(s1.locus_len_sum.sum()) > int(s.locus_len_sum.sum())
    # larger batches of pairs through the other sieves (candidates are paired up in sv_emit / k_flag_compact)
    import os
    for kind in ("global", "routed"):
        os.environ["MLST_SIEVE"] = kind
        try:
            e2, o2 = both(idx)
            e2.submit_reads(fb, fq, off, paired=True); o2.submit_reads(fb, fq, off, paired=True)
            fx.assert_stats_equal(e2.stats(), o2.stats())
        finally:
            os.environ.pop("MLST_SIEVE", None)
    with pytest.raises(MlstError):
        eng.reset_sample(); eng.submit_reads(fb[:150 * 3], fq[:150 * 3], off[:4], paired=True)       # an odd number of reads


def test_read_matching_both_strands_of_a_locus_counts_once():
    """Q3, single reads: a read whose two strands both align to one locus (a palindromic allele) has two work items there;
    the dictionary of metamlst.py:127 holds one length per (locus, QNAME)."""
    import sqlite3
    import tempfile
    from metamlst_amd.index import load_index
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(9)
    half = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=120))
    comp = half[::-1].translate(str.maketrans("ACGT", "TGCA"))
    pal = half + comp                                            # its own reverse complement
    conn = sqlite3.connect(d + "/p.db")
    synth.create_schema(conn)
    conn.execute("INSERT INTO organisms VALUES ('spP','palindromes')")
    conn.execute("INSERT INTO genes VALUES ('g0','spP')")
    for k, seq in enumerate([pal, pal[:100] + "A" + pal[101:]], start=1):
        conn.execute("INSERT INTO alleles (bacterium,gene,sequence,alignedSequence,alleleVariant) VALUES ('spP','g0',?,?,?)", (seq, seq, k))
    conn.commit(); conn.close()
    idx = load_index(d + "/p.db")
    reads = [pal[s:s + 150].encode() for s in (0, 20, 45, 90)] * 3
    fb, fq, off = synth.ragged_reads(reads, [bytes([73]) * 150] * len(reads))
    eng, orc = both(idx)
    eng.submit_reads(fb, fq, off); orc.submit_reads(fb, fq, off)
    s, so = eng.stats(), orc.stats()
    fx.assert_stats_equal(s, so)
    assert int(s.counters[5]) > int(s.counters[4])               # more work items than retained reads: both strands
    assert int(s.locus_len_sum[0]) <= 150 * len(reads)


def test_larger_multi_species_database():
    db, idx = fx.multi_species(20, 60)
    parts = []
    for k, sp in enumerate(db.species[:6]):
        g, _ = synth.make_genome(db, sp, db.profiles[sp][k % 5], size=60_000, seed=300 + k)
        parts.append(synth.sample_reads(g, 3000 + 500 * k, seed=400 + k))
    b = np.concatenate([p[0] for p in parts])
    q = np.concatenate([p[1] for p in parts])
    perm = np.random.default_rng(9).permutation(len(b))
    fb, fq, off = synth.flatten_reads(b[perm], q[perm])
    eng, orc = both(idx)
    s, _ = run_both(eng, orc, fb, fq, off)
    chosen, _ = check_pileup(eng, orc, idx, s)
    assert len({int(idx.species_id[a]) for a in chosen}) == 6


def test_error_codes_capacity_and_limits():
    from metamlst_amd.engine import MlstError
    db, idx = fx.ecoli_small(40)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 1, n_reads=8000, genome=80_000)
    # retained-read capacity too small -> MLST_E_CAPACITY reported when statistics are read
    p = default_params()
    p.max_retained_reads = 16
    eng = Engine(0, p)
    eng.load_reference(idx)
    eng.submit_reads(fb, fq, off)
    with pytest.raises(MlstError, match="capacity exceeded"):
        eng.stats()
    # pair-result arena too small
    p = default_params()
    p.max_pair_results = 4096
    eng = Engine(0, p)
    eng.load_reference(idx)
    eng.submit_reads(fb, fq, off)
    with pytest.raises(MlstError, match="capacity exceeded"):
        eng.stats()
    # a read longer than the packed format allows -> MLST_E_LIMIT, nothing submitted
    eng = Engine(0)
    eng.load_reference(idx)
    long_read = np.full(400, ord("A"), np.uint8)
    with pytest.raises(MlstError, match="longer than 320"):
        eng.submit_reads(long_read, np.full(400, 73, np.uint8), np.array([0, 400], np.uint64))
    assert eng.stats().counters[2] == 0
    # bad arguments
    with pytest.raises(MlstError):
        eng.pileup([idx.n_alleles + 5])
    with pytest.raises(MlstError):
        eng.pileup([0, 1])            # two alleles of one locus
    with pytest.raises(MlstError):
        eng.hamming_all(idx.n_loci + 1, b"ACGT")
    p = default_params()
    p.band_w = 40
    with pytest.raises(MlstError, match="band_w"):
        Engine(0, p)


def test_plain_sieve_with_global_bitmap_and_without(monkeypatch):
    """Big-database sieve variants on a small database: plain kernel with the global first-level bitmap
    (forced small so that it is selective) and with it disabled."""
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 8, n_reads=9000)
    monkeypatch.setenv("MLST_SIEVE", "global")
    for bits in ("18", "25", "0"):
        monkeypatch.setenv("MLST_GBM_BITS", bits)
        eng, orc = both(idx)
        s, _ = run_both(eng, orc, fb, fq, off)
        check_pileup(eng, orc, idx, s)


def test_device_typing_tail_equals_host_choice_and_consensus():
    """mlst_typing_enqueue / mlst_typing_fetch (allele choice + pileup + consensus queued behind pass 1) against the
    host's own allele choice (metamlst.py:133-151, 244 restated in typing.pick_alleles_fast) and mlst_consensus."""
    for db, idx, sp, row in ((*fx.ecoli_small(80), "ecoli", 4), (*fx.ecoli_small(60, indel_every=5), "ecoli", 2),
                             (*fx.multi_species(3, 25), None, 1)):
        sp = sp or db.species[1]
        fb, fq, off, _, _ = fx.isolate_reads(db, sp, row, n_reads=12000)
        eng = Engine(0)
        eng.load_reference(idx)
        for penalty in (100, 0, 7):
            eng.reset_sample()
            eng.submit_reads(fb, fq, off)
            eng.typing_enqueue(penalty=penalty)
            st, chosen, letters = eng.typing_fetch()
            fx.assert_stats_equal(st, eng.stats())
            want = pick_alleles_fast(idx, st, penalty)
            assert chosen == want
            cons = eng.consensus(sorted(want.values()))
            assert {a: bytes(v) for a, v in letters.items()} == {a: bytes(v) for a, v in cons.items()}


def test_device_allele_choice_on_crafted_ties():
    """Statistics built to tie: equal rounded averages with different allele numbers, averages on x.x5 boundaries (where
    the binary double falls on either side), exact quarters, negative penalised scores, loci without records."""
    import torch
    from metamlst_amd.typing import SampleStats
    db, idx = fx.ecoli_small(80)
    eng = Engine(0)
    eng.load_reference(idx)
    dev = torch.device("cuda", 0)
    nA, nL = idx.n_alleles, idx.n_loci
    n_sum, n_min = eng.flat_sizes()
    rng = np.random.default_rng(3)
    for trial in range(12):
        nh = np.zeros(nA, np.int64); ss = np.zeros(nA, np.int64)
        first = np.full(nL, np.iinfo(np.int64).max, np.int64)
        for l in range(nL):
            b, c = int(idx.locus_begin[l]), int(idx.locus_count[l])
            if trial % 4 == 3 and l % 3 == 0:
                continue                                  # locus without any record
            first[l] = 1000 * l + trial
            k = rng.choice(c, size=min(c, 30), replace=False)
            if trial % 3 == 0:                            # all on twentieths: p/q = (2m+1)/20
                q = rng.choice([20, 40, 100, 4, 8], size=k.size)
                m = rng.integers(1000, 1010, size=k.size)
                nh[b + k] = q
                ss[b + k] = np.where(q % 20 == 0, (2 * m + 1) * (q // 20), (2 * (m // 5) + 1) * (q // 4) + 100 * q)
            elif trial % 3 == 1:                          # identical averages: the lowest allele number must win
                q = rng.integers(1, 50, size=k.size)
                nh[b + k] = q; ss[b + k] = 250 * q
            else:                                         # unequal depths: the penalty makes scores negative
                q = rng.integers(1, 400, size=k.size)
                nh[b + k] = q; ss[b + k] = q * rng.integers(80, 300, size=k.size) + rng.integers(0, 20, size=k.size)
        flat = np.concatenate([ss, nh, np.zeros(nL, np.int64), np.zeros(n_sum - 2 * nA - nL, np.int64)])
        eng.reset_sample()
        t_flat, t_first = torch.from_numpy(flat).to(dev), torch.from_numpy(first).to(dev)
        torch.cuda.synchronize(dev)
        eng.import_stats_device(t_flat.data_ptr(), t_first.data_ptr())
        for penalty in (100, 3):
            eng.typing_enqueue(penalty=penalty)
            st, chosen, _ = eng.typing_fetch()
            assert np.array_equal(st.sum_score, ss) and np.array_equal(st.n_hits.astype(np.int64), nh)
            assert chosen == pick_alleles_fast(idx, st, penalty), (trial, penalty)


def test_streamed_shard_step_on_a_torch_stream():
    """The multi-GPU step of metamlst_amd.dist.StreamedShard -- engine on a torch stream, RCCL all-reduces queued
    between its kernels, one host synchronisation -- in a process group of one (the collectives are issued and are
    identities): same statistics, choice and consensus as the plain calls."""
    import torch
    import torch.distributed as dist
    from metamlst_amd.dist import StreamedShard
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 4, n_reads=9000)
    plain = Engine(0)
    plain.load_reference(idx)
    plain.submit_reads(fb, fq, off)
    s0 = plain.stats()
    want = pick_alleles_fast(idx, s0, 100)
    cons = plain.consensus(sorted(want.values()))
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        eng = Engine(0)
        eng.load_reference(idx)
        sh = StreamedShard(eng, torch.device("cuda", 0), force_collectives=True)
        assert sh.compact and sh.cap_cols == sh.total_cols      # the first step exchanges the fixed layout's size
        need = sum(int(idx.locus_maxlen[l]) for l in want)       # slots of the loci with a chosen allele
        caps, sbytes = [], []
        fixed_bytes = int(sh.t_all.numel()) * 8
        for k in range(5):                       # reuse across samples, as the bench loop does
            if k == 3:
                sh.cap_cols = 1024               # too small for the seven loci: the second half runs again with the full layout
            if k == 4:
                sh._set_listed([0])              # a statistics layout that misses six of the seven hit loci: exchanged again, fixed
            caps.append(sh.cap_cols)
            sh.enqueue(lambda: (eng.reset_sample(), eng.submit_reads(fb, fq, off)))
            st, chosen, letters = sh.fetch()
            sbytes.append(sh.stats_bytes_last)
            fx.assert_stats_equal(st, s0)
            assert chosen == want
            assert {a: bytes(v) for a, v in letters.items()} == {a: bytes(v) for a, v in cons.items()}
            assert sh.needs[-1] == need
        assert caps[1] == caps[2] == max(1024, min(sh.total_cols, (need * 3 // 2 + 2047) // 1024 * 1024)) and sh.repeats == 1
        # statistics: the first step in the fixed layout, then the loci that had hits (all seven here, so about the same size
        # on this one-species database; 140 of 1,050 loci on cfg3); the misfit of step 4 was seen and repaired
        assert sbytes[0] == fixed_bytes and sh.listed == tuple(range(idx.n_loci)) and sh.stats_repeats == 1
        assert sbytes[1] == (2 * idx.n_alleles + idx.n_loci + 8 + 1 + idx.n_loci) * 8
        sh.close()
        # the fixed-layout exchange stays available (MLST_COMPACT_EXCHANGE=0)
        sh = StreamedShard(eng, torch.device("cuda", 0), force_collectives=True, compact=False)
        sh.enqueue(lambda: (eng.reset_sample(), eng.submit_reads(fb, fq, off)))
        st, chosen, letters = sh.fetch()
        fx.assert_stats_equal(st, s0)
        assert chosen == want and {a: bytes(v) for a, v in letters.items()} == {a: bytes(v) for a, v in cons.items()}
        sh.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_dense_on_locus_reads_overflow_the_sieve_queue():
    """Amplicon-like data: every read comes from a locus, so nearly every seed passes the first level and the per-wave
    queue of the sieve must be drained several times per tile (early drains, synchronous rounds) -- all three sieve variants."""
    import os
    db, idx = fx.ecoli_small(80)
    rng = np.random.default_rng(21)
    seqs = [idx.sequence(int(idx.locus_begin[l]) + int(rng.integers(int(idx.locus_count[l])))) for l in range(idx.n_loci)]
    reads, quals = [], []
    for k in range(6000):
        s = seqs[k % len(seqs)]
        L = int(rng.choice([150, 150, 120, 75]))
        at = int(rng.integers(0, len(s) - L + 1))
        r = s[at:at + L].encode()
        if k % 2:
            r = r[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        reads.append(r); quals.append(bytes([40 + 33]) * L)
    fb, fq, off = synth.ragged_reads(reads, quals)
    for kind in ("lds", "global", "routed"):
        os.environ["MLST_SIEVE"] = kind
        try:
            eng, orc = both(idx)
            s, so = run_both(eng, orc, fb, fq, off)
            assert int(s.counters[4]) == 6000          # every read retained
            check_pileup(eng, orc, idx, s)
        finally:
            os.environ.pop("MLST_SIEVE", None)

def test_typing_results_survive_the_submission_of_the_next_step():
    """mlst_typing_wait / mlst_typing_fetch_waited: the results of a finished typing step are copied out AFTER the engine's
    next step (other reads) has been queued -- two pinned slots written in turn -- and equal those of mlst_typing_fetch."""
    db, idx = fx.ecoli_small(80)
    a = fx.isolate_reads(db, "ecoli", 5, n_reads=9000)[:3]
    b = fx.isolate_reads(db, "ecoli", 3, n_reads=7000, seed=synth.SEED + 9)[:3]
    eng = Engine(0)
    eng.load_reference(idx)
    want = []
    for r in (a, b):
        eng.reset_sample(); eng.submit_reads(*r); eng.typing_enqueue(penalty=100)
        want.append(eng.typing_fetch())
    eng.reset_sample(); eng.submit_reads(*a); eng.typing_enqueue(penalty=100)
    for k, r in enumerate((b, a, b)):                  # step k waited for, step k + 1 queued, step k fetched
        eng.typing_wait()
        eng.reset_sample(); eng.submit_reads(*r); eng.typing_enqueue(penalty=100)
        st, ch, let = eng.typing_fetch(waited=True)
        ws, wc, wl = want[k % 2]
        fx.assert_stats_equal(st, ws)
        assert ch == wc and {x: bytes(v) for x, v in let.items()} == {x: bytes(v) for x, v in wl.items()}
    st, ch, let = eng.typing_fetch()
    fx.assert_stats_equal(st, want[1][0])
    with pytest.raises(MlstError):
        eng.typing_wait()                              # nothing queued


@pytest.mark.parametrize("kind", ["lds", "routed"])
def test_engine_on_a_share_of_the_cus(monkeypatch, kind):
    """mlst_set_cu_partition: the engine's stream masked to a quarter, then to a seventh of the CUs, then the whole device
    again -- same statistics, items and pile-up as the oracle every time (the routed sieve sizes its producer grid by the
    share; graphs are rebuilt); mlst_busy / mlst_get_stream answer."""
    monkeypatch.setenv("MLST_SIEVE", kind)
    db, idx = fx.ecoli_small(80)
    fb, fq, off = fx.isolate_reads(db, "ecoli", 5, n_reads=30000)[:3]
    eng, _ = both(idx)
    assert eng.own_stream() != 0 and not eng.busy()
    for part, n in ((1, 4), (6, 7), (0, 1)):
        eng.set_cu_partition(part, n)
        for _ in range(3):                   # the third identical submission replays the rebuilt graph
            s, _ = run_both(eng, oracle_lib.Oracle(idx, None), fb, fq, off)
        check_pileup(eng, _fresh_oracle(idx, fb, fq, off), idx, s)
        assert not eng.busy()
    assert eng.own_stream() != 0
    with pytest.raises(MlstError):
        eng.set_cu_partition(4, 4)


def _fresh_oracle(idx, fb, fq, off):
    o = oracle_lib.Oracle(idx, None)
    o.submit_reads(fb, fq, off)
    return o


def test_routed_sieve_reused_across_submissions_of_different_sizes(monkeypatch):
    """The routed sieve keeps its candidate flags between submissions (k_flag_compact clears the words it read instead of
    a fill per submission) and hands out tiles from a counter it resets itself: a large batch, then a smaller one of other
    reads, then the large one again on ONE engine -- every pass equals the oracle, and the number of candidates equals
    that of a fresh engine (a stale flag would not change a result: k_seed looks every candidate up exactly; it would
    show as an extra candidate)."""
    monkeypatch.setenv("MLST_SIEVE", "routed")
    from metamlst_amd.engine import CNT_CANDIDATES
    db, idx = fx.ecoli_small(80)
    big = fx.isolate_reads(db, "ecoli", 5, n_reads=30000)[:3]
    small = fx.isolate_reads(db, "ecoli", 3, n_reads=7001, seed=synth.SEED + 5)[:3]
    eng, orc = both(idx)
    assert eng.sieve_info()["kind"] == "routed"
    seen = []
    for fb, fq, off in (big, small, big, small):
        orc2 = oracle_lib.Oracle(idx, None)
        s, _ = run_both(eng, orc2, fb, fq, off)
        seen.append(int(s.counters[CNT_CANDIDATES]))
    assert seen[0] == seen[2] and seen[1] == seen[3]
    fresh = Engine(0)
    fresh.load_reference(idx)
    fresh.submit_reads(*small)
    assert int(fresh.stats().counters[CNT_CANDIDATES]) == seen[1]


@pytest.mark.parametrize("kind", ["lds", "routed", "global"])
def test_read_lengths_from_36_to_300_through_every_sieve(monkeypatch, kind):
    """The sieve kernels are instantiated per row width (2 .. 20 words): batches of 36, 75, 100, 250 and 300 bp reads (whole
    batches of one length, so that the row width follows the length) through each sieve, 320 bp reads included."""
    monkeypatch.setenv("MLST_SIEVE", kind)
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][4], size=60_000)
    eng, orc = both(idx)
    for L in (36, 75, 100, 250, 300, 320):
        b, q = synth.sample_reads(g, 4000, read_len=L, seed=L)
        fb, fq, off = synth.flatten_reads(b, q)
        s, _ = run_both(eng, orc, fb, fq, off)
        if L >= 75:
            assert int(s.counters[4]) > 0
            if kind == "lds":
                check_pileup(eng, orc, idx, s)      # k_pileup_160 for rows up to 160 bases, k_pileup_320 beyond


def test_loci_longer_than_1024_columns(tmp_path):
    """Loci up to MLST_MAX_ALLELE_LEN (4,095 columns; positions are 12 bits in the seed postings, allele windows are clamped
    at both ends).  Three long loci, both row widths, against the oracle (statistics, items, pile-up, consensus)."""
    from metamlst_amd.index import load_index
    db = synth.make_db(str(tmp_path / "long.db"), {"lg": [("g0", 1500), ("g1", 3000), ("g2", 4095)]}, alleles_per_locus=12, n_profiles=6)
    idx = load_index(db.path)
    g, _ = synth.make_genome(db, "lg", db.profiles["lg"][2], size=80_000)
    eng, orc = both(idx)
    for L in (150, 300):
        b, q = synth.sample_reads(g, 12000, read_len=L, seed=L)
        fb, fq, off = synth.flatten_reads(b, q)
        s, _ = run_both(eng, orc, fb, fq, off)
        assert int(s.counters[0]) > 1000
        chosen, pc = check_pileup(eng, orc, idx, s)
        assert len(chosen) == 3 and all(int(pc[a].sum()) > 0 for a in chosen)


def test_deep_amplicon_sample():
    """1.3 M reads that all come from the seven loci (every read retained, ~160 items per k_pileup wave in three batches of
    64 lanes, ~80,000-fold columns): statistics and pile-up counts against the oracle."""
    db, idx = fx.ecoli_small(20)
    rng = np.random.default_rng(33)
    n, L = 1_300_000, 150
    seqs = [np.frombuffer(idx.sequence(int(idx.locus_begin[l]) + 3).encode(), np.uint8) for l in range(idx.n_loci)]
    which = rng.integers(0, len(seqs), n)
    b = np.empty((n, L), np.uint8)
    for l, sq in enumerate(seqs):
        rows = np.nonzero(which == l)[0]
        at = rng.integers(0, len(sq) - L + 1, rows.size)
        b[rows] = sq[at[:, None] + np.arange(L)[None, :]]
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rev = rng.random(n) < 0.5
    b[rev] = comp[b[rev][:, ::-1]]
    q = np.full((n, L), 40 + 33, np.uint8)
    fb, fq, off = synth.flatten_reads(b, q)
    p = default_params()
    p.max_retained_reads = p.max_items = 1_500_000
    p.max_pair_results = 1_500_000 * 64
    eng = Engine(0, p)
    eng.load_reference(idx)
    orc = oracle_lib.Oracle(idx, p)
    eng.submit_reads(fb, fq, off)
    orc.submit_reads(fb, fq, off)
    s, so = eng.stats(), orc.stats()
    fx.assert_stats_equal(s, so)
    assert int(s.counters[4]) == n                    # every read retained
    check_pileup(eng, orc, idx, s)


def test_replayed_typing_graph_with_foreign_copies_in_between():
    """The device-side typing tail is replayed as a hipGraph from the third identical call on.  Other users of the GPU in the
    same process (here: torch fills and copies of a few hundred MB between the samples) must not disturb it: four samples on
    one engine, every one = the host's statement of the choice and the explicit pile-up (the graphs hold
    kernel nodes only since a replay with memset / memcpy nodes faulted in exactly this kind of sequence)."""
    import torch
    db, idx = fx.ecoli_small(80)
    eng = Engine(0)
    eng.load_reference(idx)
    dev = torch.device("cuda:0")
    for k in range(4):
        fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 3 + k, n_reads=30000, seed=100 + k)
        eng.reset_sample()
        eng.submit_reads(fb, fq, off)
        eng.typing_enqueue(penalty=100)
        st, chosen, letters = eng.typing_fetch()
        want = pick_alleles_fast(idx, st, 100)
        assert chosen == want and len(want) == 7
        cons = eng.consensus(sorted(want.values()))
        assert {a: bytes(v) for a, v in letters.items()} == {a: bytes(v) for a, v in cons.items()}
        # foreign traffic: allocations, fills and copies both ways
        t = torch.empty(96 << 20, dtype=torch.int32, device=dev)
        t.fill_(k)
        h = t[: 16 << 20].cpu()
        u = torch.zeros_like(t)
        u.copy_(t)
        v = h.to(dev)
        torch.cuda.synchronize(dev)
        del t, u, v, h


def test_four_engines_four_process_groups_both_repeat_paths_over_rccl():
    """What bench.py --gpus N sets up per rank, in a group of one: FOUR engines, each with its own
    dist.new_group() (its own RCCL communicator and stream) and its own StreamedShard, steps queued round-robin as the
    pipelined loop does, and BOTH repair paths forced on every engine -- a counts buffer that is too small (the second half
    runs again in the fixed layout) and a statistics layout that misses hit loci (exchanged again, fixed).  Every step of every
    engine equals the plain calls; what is left untested before the 8-GPU run is a real peer."""
    import torch
    import torch.distributed as dist
    from metamlst_amd.dist import StreamedShard
    db, idx = fx.ecoli_small(80)
    samples = [fx.isolate_reads(db, "ecoli", 3 + k, n_reads=7000 + 500 * k)[:3] for k in range(4)]
    plain = Engine(0)
    plain.load_reference(idx)
    want = []
    for fb, fq, off in samples:
        plain.reset_sample()
        plain.submit_reads(fb, fq, off)
        s0 = plain.stats()
        ch = pick_alleles_fast(idx, s0, 100)
        want.append((s0, ch, {a: bytes(v) for a, v in plain.consensus(sorted(ch.values())).items()}))
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29543", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    shards, engines = [], []
    try:
        groups = [dist.new_group(ranks=[0]) for _ in range(4)]
        for k in range(4):
            e = Engine(0)
            e.load_reference(idx)
            engines.append(e)
            shards.append(StreamedShard(e, torch.device("cuda", 0), force_collectives=True, group=groups[k]))
        assert len({id(g) for g in groups}) == 4
        for step in range(6):
            if step == 3:
                for sh in shards:
                    sh.cap_cols = 1024           # too small for the seven loci
            if step == 4:
                for sh in shards:
                    sh._set_listed([0])          # misses six of the seven hit loci
            for k in range(4):                   # all four queued before any is fetched: their collectives are in flight together
                fb, fq, off = samples[(k + step) % 4]
                shards[k].enqueue(lambda e=engines[k], fb=fb, fq=fq, off=off: (e.reset_sample(), e.submit_reads(fb, fq, off)))
            for k in range(4):
                st, chosen, letters = shards[k].fetch()
                s0, ch, cons = want[(k + step) % 4]
                fx.assert_stats_equal(st, s0)
                assert chosen == ch and {a: bytes(v) for a, v in letters.items()} == cons, (step, k)
        assert all(sh.repeats == 1 and sh.stats_repeats == 1 for sh in shards), [(sh.repeats, sh.stats_repeats) for sh in shards]
    finally:
        for sh in shards:
            sh.close()
        for e in engines:
            e.close()
        if created:
            dist.destroy_process_group()
