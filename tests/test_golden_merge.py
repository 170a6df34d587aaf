"""metamlst-merge.py:93-494 pinned (ST tables, reports, and the --outseqformat / -j / --jgroup sequence outputs): the reference script was run on a folder of .nfo lines
(tests/golden/make_golden.py, part C); the package's merge must write identical files."""
import glob
import json
import os
import shutil
import tempfile

import pytest

import golden_util as gu
import oracle_lib
from metamlst_amd import db as mdb
from metamlst_amd.index import load_index
from metamlst_amd.merge import merge_folder

CASES = sorted(glob.glob(os.path.join(gu.GOLD, "merge", "case*")))


@pytest.mark.parametrize("case", CASES, ids=[os.path.basename(c) for c in CASES])
@pytest.mark.parametrize("cached", [False, True])
def test_reference_merged_files(case, cached):
    argv = json.load(open(os.path.join(case, "args.json")))
    z = int(argv[argv.index("-z") + 1]) if "-z" in argv else 5
    flt = argv[argv.index("--filter") + 1] if "--filter" in argv else None
    opt = lambda name: argv[argv.index(name) + 1] if name in argv else None
    meta = os.path.join(case, "meta.tsv") if opt("--meta") else None
    dbp = gu.golden_db()
    idx = load_index(dbp)
    orc = oracle_lib.Oracle(idx)

    def matcher(bacterium, gene, seq, zz):      # the stringDiff scan (merge:177-181) through the oracle
        return orc.hamming_le(idx.locus_index(bacterium, gene), seq.encode(), zz)[0] >= 0

    work = tempfile.mkdtemp()
    shutil.copy(os.path.join(case, "all.nfo"), work)
    database = mdb.metaMLST_db(dbp)
    merge_folder(work, database, matcher, z=z, filter=flt, cache=mdb.DbCache(database.conn) if cached else None,
                 meta=meta, idField=int(opt("--idField") or 0), outseqformat=opt("--outseqformat"), j=opt("-j"),
                 jgroup="--jgroup" in argv)
    want = sorted(os.listdir(os.path.join(case, "expected")))
    assert sorted(os.listdir(os.path.join(work, "merged"))) == want
    for f in want:
        assert open(os.path.join(work, "merged", f), "rb").read() == open(os.path.join(case, "expected", f), "rb").read(), f
    shutil.rmtree(work)
