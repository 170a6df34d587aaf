"""synth.metagenome_plan plants only sequence types the reference's algorithm can recover."""
import numpy as np

from metamlst_amd import synth


def test_told_apart_needs_an_interior_column():
    base = b"ACGT" * 30
    end_only = base[:-1] + b"A"                     # differs from `base` in the last column only
    inner = base[:60] + b"C" + base[61:]            # differs in column 60 (A -> C)
    seqs = {1: base, 2: end_only, 3: inner, 4: base + b"ACGT"}        # allele 4 has another length: never compared
    assert not synth._told_apart(seqs, 1, 8)        # 1 and 2 are the same once 8 columns are cut off either end
    assert not synth._told_apart(seqs, 2, 8)
    assert synth._told_apart(seqs, 3, 8)
    assert synth._told_apart({1: base, 3: inner}, 1, 8)


def test_plan_is_a_pure_function_and_plants_recoverable_types(tmp_path):
    sdb = synth.make_full_db(str(tmp_path / "p.db"), n_species=6, alleles_per_locus=40, n_profiles=12)
    plan = synth.metagenome_plan(sdb, 4)
    assert plan == synth.metagenome_plan(sdb, 4)
    assert abs(sum(fr for _, fr, _ in plan) - 1.0) < 1e-9
    import sqlite3
    conn = sqlite3.connect(sdb.path)
    for sp, _, row in plan:
        for (gene, _), allele in zip(sdb.loci[sp], sdb.profiles[sp][row]):
            seqs = {int(no): sq.encode() for no, sq in conn.execute(
                "SELECT alleleVariant, sequence FROM alleles WHERE bacterium=? AND gene=?", (sp, gene))}
            assert synth._told_apart(seqs, int(allele), 8)
    conn.close()
