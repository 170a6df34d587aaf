"""Seeded fuzz of the HIP engine against the oracle over the parameter space (GPU)."""
import sqlite3
import tempfile

import numpy as np
import pytest

import fixtures as fx
import oracle_lib
from metamlst_amd import synth
from metamlst_amd.engine import Engine, default_params
from metamlst_amd.index import load_index
from metamlst_amd.typing import consensus_from_counts, pick_alleles_fast

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = np.random.default_rng(seed)
    d = tempfile.mkdtemp()
    n_loci = int(rng.integers(1, 5))
    loci = [("g%d" % k, int(rng.choice([40, 77, 130, 256, 401, 512, 700]))) for k in range(n_loci)]
    n_all = int(rng.integers(2, 40))
    indel_every, max_div = int(rng.choice([0, 0, 3, 5])), float(rng.choice([0.03, 0.08]))
    while True:
        try:
            db = synth.make_db(d + "/f.db", {"spX": loci, "spY": loci[:1]}, alleles_per_locus=n_all, n_profiles=5, seed=seed,
                               indel_every=indel_every, max_div=max_div)
            break
        except RuntimeError:          # a short locus cannot hold that many distinct alleles within max_div: fewer alleles
            n_all = max(2, n_all // 2)
    if rng.random() < 0.5:        # sprinkle ambiguity codes / truncate some alleles
        conn = sqlite3.connect(db.path)
        for rid, seq in conn.execute("SELECT recID, sequence FROM alleles").fetchall():
            s = list(seq)
            if rng.random() < 0.3:
                for p in rng.integers(0, len(s), size=2):
                    s[p] = "NRW"[int(rng.integers(3))]
            if rng.random() < 0.15 and len(s) > 30:
                s = s[:len(s) - int(rng.integers(1, 20))]
            conn.execute("UPDATE alleles SET sequence=? WHERE recID=?", ("".join(s), rid))
        conn.commit()
        conn.close()
    idx = load_index(db.path, cluster=bool(rng.integers(2)))
    p = default_params()
    p.band_w = int(rng.integers(3, 16))
    p.gbar = int(rng.integers(0, 7))
    p.xm_field_quirk = int(rng.integers(2))
    p.gap_trigger_mm = int(rng.choice([12, 12, 4, -1]))
    p.gap_trigger_clip = int(rng.choice([8, 3, 0]))
    p.minscore = int(rng.choice([80, 40, 120]))
    p.max_xm = int(rng.choice([5, 2, 20]))
    p.min_read_len = int(rng.choice([50, 20, 100]))
    p.minqual = int(rng.choice([20, 0, 35]))
    # reads: from both species' genomes, random lengths, indels, N, random qualities
    reads, quals = [], []
    for sp in ("spX", "spY"):
        g, _ = synth.make_genome(db, sp, db.profiles[sp][0], size=6000 + 2000 * len(db.loci[sp]), seed=seed)
        n_reads = 1500 if p.gap_trigger_mm >= 0 else 500
        for k in range(n_reads):
            L = int(rng.choice([150, 150, 100, 75, 250, 301, 36, 19]))
            at = int(rng.integers(0, max(1, len(g) - L)))
            r = bytearray(g[at:at + L].tobytes())
            if rng.random() < 0.5:
                r = bytearray(synth._COMP[np.frombuffer(bytes(r), np.uint8)[::-1]].tobytes())
            q = bytearray((rng.integers(2, 42, size=len(r)) if rng.random() < 0.5 else np.full(len(r), 40)).astype(np.uint8) + 33)
            u = rng.random()
            if u < 0.15 and len(r) > 40:          # deletion in the read
                c = int(rng.integers(10, len(r) - 10)); dl = int(rng.integers(1, 5))
                del r[c:c + dl]; del q[c:c + dl]
            elif u < 0.3 and len(r) > 40:         # insertion in the read
                c = int(rng.integers(10, len(r) - 10)); ins = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 4))).astype(np.uint8))
                r[c:c] = ins; q[c:c] = bytes([40 + 33] * len(ins))
            for _ in range(int(rng.integers(0, 4))):
                pp = int(rng.integers(0, len(r))); r[pp] = b"ACGTN"[int(rng.integers(5))]
            r, q = r[:320], q[:320]
            reads.append(bytes(r)); quals.append(bytes(q))
    fb, fq, off = synth.ragged_reads(reads, quals)
    return idx, p, fb, fq, off


@pytest.mark.parametrize("seed", list(range(100, 100 + int(__import__("os").environ.get("MLST_FUZZ_N", "12")))))
def test_fuzz_engine_equals_oracle(seed):
    idx, p, fb, fq, off = random_case(seed)
    eng = Engine(0, p)
    eng.load_reference(idx)
    orc = oracle_lib.Oracle(idx, p)
    eng.submit_reads(fb, fq, off)
    orc.submit_reads(fb, fq, off)
    s = eng.stats()
    so, items_o = orc.stats(want_items=1 << 16)
    fx.assert_stats_equal(s, so)
    assert np.array_equal(fx.sorted_items(eng.items(1 << 16)), fx.sorted_items(items_o))
    chosen = sorted(pick_alleles_fast(idx, s, 100).values())
    if chosen:
        pc, po = eng.pileup(chosen), orc.pileup(chosen)
        cons = eng.consensus(chosen, mincov=2)
        for a in chosen:
            assert np.array_equal(pc[a], po[a]), "pileup differs for allele %d" % a
            assert cons[a].decode() == "".join(consensus_from_counts(po[a], mincov=2))
    for locus in range(idx.n_loci):
        q = idx.sequence(int(idx.locus_begin[locus]))[3:].encode()
        assert np.array_equal(eng.hamming_all(locus, q), orc.hamming_all(locus, q))
