"""metamlst-index.py:90-217 pinned: the reference script built a database from FASTA + typings files
(tests/golden/make_golden.py, part D); metamlst_amd/dbbuild.py must produce the same tables and log."""
import json
import os
import shutil
import sqlite3
import tempfile

import golden_util as gu
from metamlst_amd import dbbuild

BASE = os.path.join(gu.GOLD, "dbbuild")


def dump_tables(path):
    conn = sqlite3.connect(path)
    out = {t: [list(r) for r in conn.execute("SELECT * FROM %s ORDER BY rowid" % t)] for t in ("organisms", "genes", "alleles", "profiles")}
    conn.close()
    return out


def test_reference_tables_and_logfile():
    work = tempfile.mkdtemp()
    for fn in ("a.fasta", "b.fasta", "typ.txt", "typ2.txt"):
        shutil.copy(os.path.join(BASE, fn), work)
    db = os.path.join(work, "new.db")
    conn = dbbuild.open_db(db)
    rep = dbbuild.add_sequences(conn, [work + "/a.fasta", work + "/b.fasta"])
    dbbuild.add_typings(conn, [work + "/typ.txt", work + "/typ2.txt"], logfile=work + "/metamlst_logfile.log")
    conn.close()
    assert dump_tables(db) == json.load(open(os.path.join(BASE, "expected_tables.json")))
    assert open(work + "/metamlst_logfile.log", "rb").read() == open(os.path.join(BASE, "expected_logfile.log"), "rb").read()
    assert "spC_g1" in rep[work + "/a.fasta"]["skipped"] and "spC_g1_2" in rep[work + "/b.fasta"]["skipped"]
    # second run on the same database (alleles already present, profiles replaced)
    conn = dbbuild.open_db(db)
    dbbuild.add_sequences(conn, [work + "/a.fasta"])
    dbbuild.add_typings(conn, [work + "/typ.txt"], logfile=work + "/metamlst_logfile.log")
    conn.close()
    assert dump_tables(db) == json.load(open(os.path.join(BASE, "expected_tables_second_run.json")))
    # the built database loads into the engine's index form and dumps back to FASTA
    from metamlst_amd.index import load_index
    idx = load_index(db)
    assert idx.n_loci == 4 and idx.n_alleles == 13
    conn = dbbuild.open_db(db)
    assert dbbuild.dump_db_to_fasta(conn, work + "/out.fa") == 13
    assert dbbuild.dump_db_to_fasta(conn, work + "/out2.fa", "spE") == 1
    conn.close()
    assert sorted(i for i, _ in dbbuild.read_fasta(work + "/out.fa")) == sorted(idx.label(a) for a in range(idx.n_alleles))
