"""Known-answer and property tests of the oracle's ALIGN (the engine's specification of row a1).
bowtie2 itself is not available, so these pin the documented local-mode scoring rules
(mlst_policy.h) on cases small enough to verify by hand."""
import numpy as np

import fixtures as fx
import oracle_lib
from metamlst_amd.engine import default_params


def setup():
    db, idx = fx.ecoli_small(20, indel_every=0, seed=31)
    return idx, oracle_lib.Oracle(idx)


def rc(s):
    return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))


def test_perfect_read_scores_two_per_base_both_strands():
    idx, o = setup()
    seq = idx.sequence(3).encode()
    r = seq[100:250]
    q = b"I" * 150
    a = o.align_one(r, q, 3, 0, 100)
    assert (a["score"], a["xm"], a["xo"], a["mm_total"]) == (300, 0, 0, 0) and len(a["cols"]) == 150
    b = o.align_one(rc(r), q, 3, 1, 100)
    assert (b["score"], b["xm"], b["xo"]) == (300, 0, 0) and b["cols"] == a["cols"]


def test_quality_aware_mismatch_penalty():
    idx, o = setup()
    seq = idx.sequence(3).encode()
    for phred, pen in ((40, 6), (30, 5), (20, 4), (15, 3), (2, 2), (0, 2)):
        r = bytearray(seq[100:250])
        r[75] = ord("A") if r[75] != ord("A") else ord("C")
        q = bytearray(b"I" * 150)
        q[75] = 33 + phred
        a = o.align_one(bytes(r), bytes(q), 3, 0, 100)
        assert (a["score"], a["xm"]) == (2 * 149 - pen, 1), phred


def test_soft_clip_of_a_bad_end_and_of_the_allele_edge():
    idx, o = setup()
    seq = idx.sequence(3).encode()
    r = bytearray(seq[100:250])
    for p in (2, 4, 6):                           # three mismatches in the first 7 bases: clipping them wins
        r[p] = ord("A") if r[p] != ord("A") else ord("C")
    a = o.align_one(bytes(r), b"I" * 150, 3, 0, 100)
    assert a["score"] == 2 * 143 and a["xm"] == 0 and a["cols"][0] == (7, 107) and a["mm_total"] == 3
    # read hanging 30 bases over the start of the allele: those bases are unaligned (soft clipped)
    r2 = b"ACGT" * 7 + b"AC" + seq[:120]
    a2 = o.align_one(r2, b"I" * 150, 3, 0, -30)
    assert a2["score"] == 240 and a2["cols"][0] == (30, 0)


def test_n_costs_one_and_counts_as_mismatch():
    idx, o = setup()
    seq = idx.sequence(3).encode()
    r = bytearray(seq[100:250])
    r[60] = ord("N")
    a = o.align_one(bytes(r), b"I" * 150, 3, 0, 100)
    assert (a["score"], a["xm"]) == (2 * 149 - 1, 1)


def test_banded_sw_recovers_an_indel_and_beats_ungapped():
    idx, o = setup()
    seq = idx.sequence(3).encode()
    r_del = seq[100:170] + seq[173:253]           # read lacks 3 allele bases: a read gap of length 3
    u = o.align_one(r_del, b"I" * 150, 3, 0, 100, mode=1)
    g = o.align_one(r_del, b"I" * 150, 3, 0, 100, mode=2)
    assert g["score"] == 300 - (5 + 3 * 3) and g["xo"] == 1 and g["xm"] == 0
    assert u["score"] < g["score"] and u["mm_total"] > 12
    p = o.align_one(r_del, b"I" * 150, 3, 0, 100, mode=0)     # policy: trigger fires -> banded result
    assert p == g
    r_ins = seq[100:170] + b"GG" + seq[170:248]   # read has 2 extra bases: a reference gap of length 2
    g2 = o.align_one(r_ins, b"I" * 150, 3, 0, 100, mode=2)
    assert g2["score"] == 2 * 148 - (5 + 3 * 2) and g2["xo"] == 1
    missing = sorted(set(range(150)) - {i for i, _ in g2["cols"]})     # inserted bases are not piled up
    assert len(missing) == 2 and missing[1] == missing[0] + 1 and 64 <= missing[0] <= 76


def test_gbar_forbids_gaps_near_the_read_ends():
    idx, o = setup()
    seq = idx.sequence(3).encode()
    r = seq[100:102] + seq[104:252]               # deletion after read position 2 (< gbar 4)
    g = o.align_one(r, b"I" * 150, 3, 0, 100, mode=2)
    assert g["xo"] == 0 and g["score"] >= 2 * 148 # no gap this close to the end: the band's shifted diagonal is used instead
    r2 = seq[100:110] + seq[113:253]              # the same 3-base deletion at read position 10 (>= gbar) is a gap
    assert o.align_one(r2, b"I" * 150, 3, 0, 100, mode=2)["xo"] == 1


def test_banded_never_below_ungapped_random():
    idx, o = setup()
    rng = np.random.default_rng(5)
    seq = idx.sequence(7).encode()
    for _ in range(60):
        at = int(rng.integers(0, len(seq) - 120))
        r = bytearray(seq[at:at + 120])
        for p in rng.integers(0, 120, size=int(rng.integers(0, 12))):
            r[p] = b"ACGT"[int(rng.integers(4))]
        q = bytes(rng.integers(2, 42, size=120).astype(np.uint8) + 33)
        d = at + int(rng.integers(-2, 3))
        u = o.align_one(bytes(r), q, 7, 0, d, mode=1)
        g = o.align_one(bytes(r), q, 7, 0, d, mode=2)
        assert g["score"] >= u["score"]
        if d == at and u["mm_total"] == 0:
            assert g["score"] == u["score"] == 240
