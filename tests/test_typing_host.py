"""End-to-end typing on the CPU (oracle as the engine stand-in): planted ST recovered through
.nfo + merge; planted SNPs become an accepted new allele; thin coverage is dropped by the
accuracy gate; fast and full host paths agree.  These are the policy-independent known answers
of SURVEY.md 8(c)."""
import os
import tempfile

import numpy as np

import fixtures as fx
import oracle_lib
from metamlst_amd import db as mdb
from metamlst_amd import synth
from metamlst_amd.merge import merge_folder
from metamlst_amd.typing import TypingArgs, compile_cel, pick_alleles, pick_alleles_fast, type_sample


def run(idx, db, fb, fq, off, out, name="iso1"):
    orc = oracle_lib.Oracle(idx)
    orc.submit_reads(fb, fq, off)
    st = orc.stats()
    database = mdb.metaMLST_db(db.path)
    res = type_sample(idx, st, orc.pileup, database, name, TypingArgs(), out_dir=out)
    return orc, st, res, database


def matcher_for(orc, idx):
    return lambda b, g, s, z: orc.hamming_le(idx.locus_index(b, g), s.encode(), z)[0] >= 0


def test_error_free_isolate_gives_planted_st():
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 9, n_reads=20000, err=0.0)
    out = tempfile.mkdtemp()
    orc, st, res, database = run(idx, db, fb, fq, off, out)
    assert res[0].written and os.path.exists(out + "/iso1.nfo")
    tables = merge_folder(out, database, matcher_for(orc, idx))
    assert tables["ecoli"]["isolates"] == [(10, 100.0, "iso1")]
    rep = open(out + "/merged/ecoli_report.txt").read().splitlines()
    assert rep[1].split("\t") == ["10", "100.0", "iso1"]


def test_planted_snps_become_accepted_new_allele():
    db, idx = fx.ecoli_small(80)
    seq = synth.allele_sequence(db.path, "ecoli", "icd", int(db.profiles["ecoli"][4][3]))
    mut = [(200, "A" if seq[200] != "A" else "C"), (333, "G" if seq[333] != "G" else "T")]
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 4, n_reads=20000, mutate={"icd": mut})
    out = tempfile.mkdtemp()
    orc, st, res, database = run(idx, db, fb, fq, off, out)
    icd = [r for r in res[0].loci_report if r["locus"] == "icd"][0]
    assert icd["snps"] >= 2 and icd["notes"] == "NEW" and res[0].newProfile == 1
    tables = merge_folder(out, database, matcher_for(orc, idx), z=5)["ecoli"]
    assert tables["isolates"][0][0] == 100001                     # new ST minted from 100001 (merge:134,229)
    prof = tables["encounteredProfiles"][100001]
    assert prof[0]["icd"] == ("100001", 1) and prof[2] == 1       # new allele number 100001, accepted (blue)
    # with z=0 the same allele is rejected
    out2 = tempfile.mkdtemp()
    open(out2 + "/iso1.nfo", "w", newline="").write(open(out + "/iso1.nfo", newline="").read())
    t2 = merge_folder(out2, database, matcher_for(orc, idx), z=0)["ecoli"]
    assert t2["isolates"] == [] and t2["encounteredProfiles"][100001][2] == 3


def test_thin_coverage_is_dropped_by_accuracy_gate():
    db, idx = fx.ecoli_small(80)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 2, n_reads=1500, genome=200_000)    # ~1x
    out = tempfile.mkdtemp()
    orc, st, res, database = run(idx, db, fb, fq, off, out)
    assert res and all(not r.written for r in res)
    assert not os.path.exists(out + "/iso1.nfo")


def test_fast_choice_equals_reference_choice():
    for row in (0, 3, 8):
        db, idx = fx.ecoli_small(80)
        fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", row, n_reads=6000, seed=row + 1)
        orc = oracle_lib.Oracle(idx)
        orc.submit_reads(fb, fq, off)
        st = orc.stats()
        cel = compile_cel(idx, st, 100)
        slow = {g: int(k) for g, k in pick_alleles(cel["ecoli"], "ecoli")}
        fast = {idx.loci[l][1]: int(idx.allele_no[a]) for l, a in pick_alleles_fast(idx, st, 100).items()}
        assert slow == fast


def test_rescue_policy_and_always_banded_agree_on_indel_free_database():
    """On a database without indels the gap-trigger policy and exhaustive banded SW give the same
    statistics (the policy only skips work that cannot change the result there)."""
    db, idx = fx.ecoli_small(10)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 1, n_reads=1500, genome=40_000)
    from metamlst_amd.engine import default_params
    p = default_params()
    p.gap_trigger_mm = -1
    a = oracle_lib.Oracle(idx)
    b = oracle_lib.Oracle(idx, p)
    a.submit_reads(fb, fq, off)
    b.submit_reads(fb, fq, off)
    sa, sb = a.stats(), b.stats()
    assert np.array_equal(sa.n_hits, sb.n_hits)
    assert (sb.sum_score >= sa.sum_score).all() and (sb.sum_score - sa.sum_score).sum() <= 0.001 * sa.sum_score.sum()
