"""BASELINE configs[4] carried to the artefact: "paired-end 2 x 150 bp at 5x locus coverage -- stresses
consensus gap-fill + closest-allele tie-break".

A paired sample at ~5x goes through the whole path -- pass 1, allele choice on the device (metamlst.py:133-151, 244), pile-up,
majority consensus, gap-fill and SNP count (metaMLST_functions.py:260-276), the .nfo line (metamlst.py:284-285), allele match
and ST call (metamlst-merge.py:144-240) -- and every stage is compared with the oracle + host statement of the same lines:

* the gap-fill count of every locus (CI) and the filled, lower-cased bases;
* a tie: a column of a planted allele that the sample left without a read (at 5x, e^-5 of the columns) gets a twin allele in
  the database -- lower allele number, different in that column only: the twins collect the same records, metamlst.py:244
  takes the lower number although the genome carries the other, the hole is filled from the twin
  (metaMLST_functions.py:265-268), and the ST that comes out is not the planted one -- from the reference's lines as from
  the engine's;
* the ST.
"""
import os
import sqlite3
import tempfile

import numpy as np
import pytest

import oracle_lib
from metamlst_amd import db as mdb
from metamlst_amd import synth
from metamlst_amd.engine import Engine
from metamlst_amd.index import load_index
from metamlst_amd.merge import EngineMatcher, SpeciesSession, parse_nfo_line
from metamlst_amd.typing import TypingArgs, compile_cel, pick_alleles_fast, type_sample

pytestmark = pytest.mark.gpu


def _plant_a_twin(db, idx, orc_stats, pile, planted_row):
    """A column of a planted allele that the 5x sample left uncovered (e^-5 of them are) -> the database gets a TWIN of that
    allele under a lower allele number, different in that column only.  -> (gene, column, twin number, planted number)"""
    chosen = pick_alleles_fast(idx, orc_stats, 100)
    if any((pile[a].sum(axis=1) == 0).mean() > 0.08 for a in chosen.values()):
        raise AssertionError("a locus of this sample is too thin for the accuracy gate (metamlst.py:262)")
    for k, (gene, _) in enumerate(db.loci["ecoli"]):
        planted = int(db.profiles["ecoli"][planted_row][k])
        a = chosen[idx.locus_index("ecoli", gene)]
        if planted < 2 or int(idx.allele_no[a]) != planted:
            continue
        cov = pile[a].sum(axis=1)
        holes = [int(c) for c in np.nonzero(cov == 0)[0] if 20 < c < len(cov) - 20]
        if not holes:
            continue
        c = holes[0]
        seq = idx.sequence(a)
        twin = seq[:c] + ("A" if seq[c] != "A" else "C") + seq[c + 1:]
        conn = sqlite3.connect(db.path)
        conn.execute("UPDATE alleles SET sequence=?, alignedSequence=? WHERE bacterium='ecoli' AND gene=? AND alleleVariant=1", (twin, twin, gene))
        conn.commit()
        conn.close()
        return gene, c, 1, planted
    raise AssertionError("no planted allele with an uncovered interior column in this sample")


def test_cfg5_paired_sample_to_its_nfo_line_and_st():
    with tempfile.TemporaryDirectory() as d:
        db = synth.make_ecoli_db(os.path.join(d, "c5.db"), alleles_per_locus=40, n_profiles=25)
        row = 3
        g, starts = synth.make_genome(db, "ecoli", db.profiles["ecoli"][row], size=100_000, seed=31)
        targs = TypingArgs()
        idx0 = load_index(db.path)
        for seed in range(32, 72):                   # the first 5x sample that leaves an interior column of a planted allele uncovered
            b, q = synth.sample_pairs(g, n_pairs=int(100_000 * 5 / 300), seed=seed)    # 2 x 150 at ~5x
            fb, fq, off = synth.flatten_reads(b, q)
            orc0 = oracle_lib.Oracle(idx0)
            orc0.submit_reads(fb, fq, off, paired=True)
            so0 = orc0.stats()
            try:
                gene, m, twin_no, planted_no = _plant_a_twin(db, idx0, so0, orc0.pileup(sorted(pick_alleles_fast(idx0, so0, 100).values())), row)
                break
            except AssertionError:
                continue
        else:
            pytest.fail("forty 5x samples without a usable uncovered interior column")
        idx = load_index(db.path)                    # the database with the twin
        database = mdb.metaMLST_db(db.path)
        # ---- oracle + host statement of metamlst.py:133-289
        orc = oracle_lib.Oracle(idx)
        orc.submit_reads(fb, fq, off, paired=True)
        so = orc.stats()
        want = type_sample(idx, so, orc.pileup, database, "s5", targs, out_dir=None)
        assert len(want) == 1 and want[0].written, ("5x must pass the accuracy gate (metamlst.py:262)", want[0].loci_report)
        # ---- engine: the device tail
        eng = Engine(0)
        eng.load_reference(idx)
        eng.submit_reads(fb, fq, off, paired=True)
        eng.typing_enqueue(penalty=targs.penalty)
        st, chosen, letters = eng.typing_fetch()
        assert np.array_equal(st.sum_score, so.sum_score) and np.array_equal(st.n_hits, so.n_hits) and np.array_equal(st.locus_len_sum, so.locus_len_sum)
        got = type_sample(idx, st, None, database, "s5", targs, out_dir=None, typed=(chosen, letters))
        assert got[0].nfo_line == want[0].nfo_line                                      # byte for byte (Q6 order, float quirks)
        # ---- the tie: no read tells the twins apart, both collect the same records; the lower number is chosen (metamlst.py:244)
        # although the genome carries the other one
        cel = compile_cel(idx, st, targs.penalty)["ecoli"][gene]
        top = max(v[2] for v in cel.values())
        tied = sorted(int(k) for k, v in cel.items() if v[2] == top)
        assert tied[:2] == [twin_no, planted_no] and cel[str(twin_no)] == cel[str(planted_no)], tied
        l0 = idx.locus_index("ecoli", gene)
        assert int(idx.allele_no[chosen[l0]]) == twin_no == int(idx.allele_no[pick_alleles_fast(idx, so, targs.penalty)[l0]])
        # ---- gap-fill: holes exist at 5x, every one filled from the chosen allele in lower case (metaMLST_functions.py:265-268)
        holes = {r["locus"]: int(r["ns"]) for r in got[0].loci_report}
        assert holes == {r["locus"]: int(r["ns"]) for r in want[0].loci_report} and sum(holes.values()) > 0
        organism, (line, sample) = parse_nfo_line(got[0].nfo_line)
        raw = dict(x.split("::")[:2] for x in got[0].nfo_line.split()[2:])
        for label, seq in raw.items():
            if not seq:
                continue
            ref = dict(got[0].chosen)[label]
            assert len(seq) == len(ref) and sum(c.islower() for c in seq) == holes[label.split("_")[1]]
            assert all(c.upper() == r for c, r in zip(seq, ref) if c.islower())
        # the twins' column is a hole, filled from the CHOSEN allele (the twin) in lower case: no SNP there, so the planted allele is
        # not recoverable -- by the reference either (its merge step upper-cases the line, metamlst-merge.py:107)
        lab1 = "ecoli_%s_%d" % (gene, twin_no)
        col = dict(got[0].chosen)[lab1][m]
        rep0 = [r for r in got[0].loci_report if r["locus"] == gene][0]
        assert int(rep0["ns"]) >= 1
        if raw[lab1]:
            assert raw[lab1][m] == col.lower()
        # ---- allele match + ST (metamlst-merge.py:144-240)
        sess = SpeciesSession(database, "ecoli", 5, EngineMatcher(eng, idx), mdb.DbCache(database.conn))
        st_called = sess.add_sample(line, sample)
        sess_o = SpeciesSession(database, "ecoli", 5, EngineMatcher(eng, idx), mdb.DbCache(database.conn))
        st_oracle = sess_o.add_sample(*parse_nfo_line(want[0].nfo_line)[1])
        assert st_called == st_oracle
        # what comes out is the profile with allele 1 at that locus (if the database has it) -- in any case the same call from
        # the oracle's line and the engine's, and not the planted ST
        assert st_called != row + 1 or st_called is None
        eng.close()
        database.closeConnection()
